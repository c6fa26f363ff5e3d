// codec5_kernels.hpp -- gfx950 kernels for the 5-letter {A,C,G,T/U,N} codec.
//
// Replaces the loops of n_to_bits2_{lut,pext} (reference src/n_to_bits2.rs:37-74,
// :118-189) and bits_to_n2_{lut,pdep} (:78-107, :196-268): 3 nt -> a + 5b + 25c
// (7 bits), 9 triplets (27 nt) per u64, bit 63 always 0.
//
// 27-nt words do not line up with power-of-two vectors (64 B of ASCII = 2.37
// words), which is where LDS earns its keep on this path: a WAVE owns 64 words =
// 1728 B of ASCII = 108 x 16 B, moves them between HBM and its private LDS slab
// with coalesced 16-B accesses, and each lane works on one whole word out of LDS.
// Slabs are per wave, so there is no workgroup barrier -- only the wave-level
// ordering fence -- and workgroups can be a single wave (the 2-bit codec's lesson:
// many tiny workgroups stream best).  Same cache-policy bits as the 2-bit kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec2_kernels.hpp"

namespace cnt {

// ---- per-byte code tables ---------------------------------------------------
// default mode: 8-entry table on the low 3 ASCII bits, the reference's own
// trick (n_to_bits2.rs:127-136): A(1)->0 C(3)->1 T(4)->2 U(5)->2 N(6)->4 G(7)->3,
// bytes >= 0x80 -> 0 (pshufb zeroes them, :151).  One v_perm_b32 per 4 bytes.
__device__ __forceinline__ uint32_t code5_fast(uint32_t x) {
    const uint32_t lut_lo = 0x01000000u;  // k=0:0 1:A=0 2:0 3:C=1
    const uint32_t lut_hi = 0x03040202u;  // k=4:T=2 5:U=2 6:N=4 7:G=3
    uint32_t c = __builtin_amdgcn_perm(lut_hi, lut_lo, x & 0x07070707u);
    uint32_t hi = x & 0x80808080u;                       // bytes >= 0x80 -> 0
    return c & ~((hi >> 5) | (hi >> 6) | (hi >> 7));     // clear bits 2..0 of those bytes
}

// CNT_STRICT_LUT: BYTE_LUT semantics (n_to_bits2.rs:8-23): anything that is not
// one of ACGTUNacgtun encodes as 0.
__device__ __forceinline__ uint32_t code5_strict(uint32_t x) {
    const uint32_t exp_lo = 0x43FF41FFu;  // k=0:FF 1:'A' 2:FF 3:'C'
    const uint32_t exp_hi = 0x474E5554u;  // k=4:'T' 5:'U' 6:'N' 7:'G'
    uint32_t expect = __builtin_amdgcn_perm(exp_hi, exp_lo, x & 0x07070707u);
    uint32_t z = (x & 0xDFDFDFDFu) ^ expect;
    uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;  // 0x80 per invalid byte
    const uint32_t lut_lo = 0x01000000u, lut_hi = 0x03040202u;
    uint32_t c = __builtin_amdgcn_perm(lut_hi, lut_lo, x & 0x07070707u);
    return c & ~((nz >> 5) | (nz >> 6) | (nz >> 7));
}

// the two halves of code5_fast, for callers that can skip the second one for a whole wave when
// no lane saw a byte >= 0x80 (always, on text)
__device__ __forceinline__ uint32_t code5_table(uint32_t x) {
    return __builtin_amdgcn_perm(0x03040202u, 0x01000000u, x & 0x07070707u);
}
__device__ __forceinline__ uint32_t code5_clear_high(uint32_t x, uint32_t c) {
    const uint32_t hi = x & 0x80808080u;
    return c & ~((hi >> 5) | (hi >> 6) | (hi >> 7));
}

template <bool STRICT>
__device__ __forceinline__ uint32_t code5(uint32_t x) {
    if constexpr (STRICT) return code5_strict(x);
    else return code5_fast(x);
}

// 27 codes (one per byte, 7 dwords, each 0..4) -> one packed word.  A triplet's value
// a + 5b + 25c is a byte dot product, so v_dot4_u32_u8 does it in one instruction when the
// triplet sits inside a dword and in two (chained through the accumulator) when it straddles:
// 13 dot products for the 9 fields.  Byte 27 (top byte of c[6]) has weight 0 everywhere and may
// hold anything.
__device__ __forceinline__ uint64_t pack27(const uint32_t (&c)[7]) {
    constexpr uint32_t W012 = 0x00190501u;  // bytes 0,1,2 of the dword  x (1, 5, 25)
    constexpr uint32_t W3 = 0x01000000u;    // byte 3 x 1 ...
    constexpr uint32_t W01 = 0x00001905u;   // ... continued by bytes 0,1 of the next dword x (5, 25)
    constexpr uint32_t W23 = 0x05010000u;   // bytes 2,3 x (1, 5) ...
    constexpr uint32_t W0 = 0x00000019u;    // ... continued by byte 0 of the next dword x 25
    constexpr uint32_t W123 = 0x19050100u;  // bytes 1,2,3 x (1, 5, 25)
#define CNT_DOT(A, W, ACC) __builtin_amdgcn_udot4((A), (W), (ACC), false)
    const uint32_t v0 = CNT_DOT(c[0], W012, 0u);
    const uint32_t v1 = CNT_DOT(c[1], W01, CNT_DOT(c[0], W3, 0u));
    const uint32_t v2 = CNT_DOT(c[2], W0, CNT_DOT(c[1], W23, 0u));
    const uint32_t v3 = CNT_DOT(c[2], W123, 0u);
    const uint32_t v4 = CNT_DOT(c[3], W012, 0u);
    const uint32_t v5 = CNT_DOT(c[4], W01, CNT_DOT(c[3], W3, 0u));
    const uint32_t v6 = CNT_DOT(c[5], W0, CNT_DOT(c[4], W23, 0u));
    const uint32_t v7 = CNT_DOT(c[5], W123, 0u);
    const uint32_t v8 = CNT_DOT(c[6], W012, 0u);
#undef CNT_DOT
    const uint32_t lo = v0 | (v1 << 7) | (v2 << 14) | (v3 << 21) | (v4 << 28);
    const uint32_t hi = (v4 >> 4) | (v5 << 3) | (v6 << 10) | (v7 << 17) | (v8 << 24);
    return ((uint64_t)hi << 32) | lo;
}

// one 7-bit value -> 3 code digits (a,b,c) as bytes 0..2 of a dword; c clamped
// to 4 for the values 125..127 no encoder produces (reference reads past its
// 5-entry LUT there; the oracle defines 'N').
__device__ __forceinline__ uint32_t digits3(uint32_t v) {
    // all products fit 24 bits: v_mul_u32_u24 / v_mad_u32_u24 are full rate, v_mul_lo_u32 is not
    uint32_t c = __umul24(v, 41u) >> 10;   // v/25 for v < 128 (checked in tests/test_bit_tricks.py)
    uint32_t r = v - __umul24(c, 25u);
    uint32_t b = __umul24(r, 13u) >> 6;    // r/5 for r < 69
    uint32_t a = r - __umul24(b, 5u);
    c = c > 4u ? 4u : c;
    return a | (b << 8) | (c << 16);
}

// the nine 7-bit fields of a packed word held as two dwords, with 32-bit ops only:
// fields 0..3 live in lo[0..27], field 4 straddles (lo[28..31], hi[0..2]), fields 5..8 in hi[3..30]
template <int K>
__device__ __forceinline__ uint32_t field7(uint32_t lo, uint32_t hi) {
    if constexpr (K < 4) return __builtin_amdgcn_ubfe(lo, 7 * K, 7);
    else if constexpr (K == 4) return __builtin_amdgcn_alignbit(hi, lo, 28) & 0x7Fu;
    else return __builtin_amdgcn_ubfe(hi, 7 * K - 32, 7);
}

__device__ __forceinline__ uint32_t letters5(uint32_t codes /* 4 code bytes, each 0..4 */) {
    // v_perm_b32 as an 8-entry table: 0..3 -> "ACTG", 4 -> 'N'
    return __builtin_amdgcn_perm(0x4E4E4E4Eu /* 'N' x4 = entries 4..7 */, 0x47544341u, codes);
}

// ---- decode arithmetic on field PAIRS ------------------------------------------------------
// Two 7-bit values in the 16-bit halves of a dword -> their digits with packed 16-bit math
// (v_pk_mul_lo_u16 / v_pk_lshrrev_b16 / v_pk_mad_u16: 6 instructions for both).  c may come out
// as 5 for the values 125..127 no encoder produces; the letter table maps 4..7 to 'N', which is
// what the clamp in digits3 does.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
struct PairLetters {
    uint32_t t;  // letters of (a_even, b_even, a_odd, b_odd)
    uint32_t c;  // letters of (c_even, -, c_odd, -)
};
__device__ __forceinline__ void digits_pair(uint32_t x, uint32_t& ab, uint32_t& cc) {
    const u16x2 v = __builtin_bit_cast(u16x2, x);
    const u16x2 c = (v * (unsigned short)41) >> (unsigned short)10;  // v/25, v < 128
    const u16x2 r = v - c * (unsigned short)25;
    const u16x2 b = (r * (unsigned short)13) >> (unsigned short)6;   // r/5, r < 69
    const u16x2 a = r - b * (unsigned short)5;
    ab = __builtin_bit_cast(uint32_t, a) | (__builtin_bit_cast(uint32_t, b) << 8);  // bytes a_e, b_e, a_o, b_o
    cc = __builtin_bit_cast(uint32_t, c);                                            // bytes c_e, 0, c_o, 0
}
__device__ __forceinline__ PairLetters letters_pair(uint32_t x) {
    uint32_t ab, cc;
    digits_pair(x, ab, cc);
    return PairLetters{letters5(ab), letters5(cc)};
}

// One packed word (lo, hi) -> its 27 letters as 7 dwords (top byte of b[6] = 0).  v_perm_b32
// with constant selectors weaves the pair results into string order: per pair the string bytes
// are t0 t1 c0 t2 | t3 c2.
__device__ __forceinline__ void decode27(uint32_t lo, uint32_t hi, uint32_t (&b)[7]) {
    const uint32_t f4 = __builtin_amdgcn_alignbit(hi, lo, 28) & 0x7Fu;  // the field that straddles the dwords
    const uint32_t x0 = __builtin_amdgcn_ubfe(lo, 0, 7) | (__builtin_amdgcn_ubfe(lo, 7, 7) << 16);
    const uint32_t x1 = __builtin_amdgcn_ubfe(lo, 14, 7) | (__builtin_amdgcn_ubfe(lo, 21, 7) << 16);
    const uint32_t x2 = f4 | (__builtin_amdgcn_ubfe(hi, 3, 7) << 16);
    const uint32_t x3 = __builtin_amdgcn_ubfe(hi, 10, 7) | (__builtin_amdgcn_ubfe(hi, 17, 7) << 16);
    const uint32_t x4 = __builtin_amdgcn_ubfe(hi, 24, 7);
    const PairLetters p0 = letters_pair(x0), p1 = letters_pair(x1), p2 = letters_pair(x2), p3 = letters_pair(x3);
    // perm(hi_src, lo_src, sel): selector bytes 0..3 pick from lo_src, 4..7 from hi_src, 0x0C gives 0
    constexpr uint32_t kHead = 0x02040100u;  // t0 t1 c0 t2        (string bytes 0..3 of a pair)
    constexpr uint32_t kRest = 0x0C0C0603u;  // t3 c2 0 0          (string bytes 4..5)
    constexpr uint32_t kJoin = 0x05040100u;  // rest0 rest1 | next pair's t0 t1
    constexpr uint32_t kTail = 0x06030204u;  // c0 t2 t3 c2        (string bytes 2..5)
    b[0] = __builtin_amdgcn_perm(p0.c, p0.t, kHead);
    b[1] = __builtin_amdgcn_perm(p1.t, __builtin_amdgcn_perm(p0.c, p0.t, kRest), kJoin);
    b[2] = __builtin_amdgcn_perm(p1.c, p1.t, kTail);
    b[3] = __builtin_amdgcn_perm(p2.c, p2.t, kHead);
    b[4] = __builtin_amdgcn_perm(p3.t, __builtin_amdgcn_perm(p2.c, p2.t, kRest), kJoin);
    b[5] = __builtin_amdgcn_perm(p3.c, p3.t, kTail);
    uint32_t ab, cc;
    digits_pair(x4, ab, cc);  // the ninth field alone: selector bytes a, b, c and 0x0C (-> 0x00)
    b[6] = __builtin_amdgcn_perm(0x4E4E4E4Eu, 0x47544341u, ab | (cc << 16) | 0x0C000000u);
}

// One packed word from byte loads, any alignment, any length (missing digits = 0, n_to_bits2.rs:58-70); `lut` = BYTE_LUT
// semantics for THIS word (CNT_STRICT_LUT everywhere; CNT_TAIL_LUT from the word where n_to_bits2_pext hands over to
// n_to_bits2_lut, n_to_bits2.rs:120,179-185).
__device__ __forceinline__ uint64_t encode2_word_bytes(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t w, bool lut) {
    const uint64_t i0 = w * 27;
    const int m = (n_len - i0) < 27 ? (int)(n_len - i0) : 27;
    uint32_t c[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * d + j;
            if (k < 27 && k < m) x |= (uint32_t)n[i0 + k] << (8 * j);
        }
        c[d] = lut ? code5_strict(x) : code5_fast(x);  // unloaded bytes are 0 -> code 0
    }
    return pack27(c);
}

// the same, also counting the word's bytes outside ACGTUNacgtun into `bad` (bytes that do not exist count as 'A')
__device__ __forceinline__ uint64_t encode2_word_bytes_checked(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t w, bool lut, uint32_t& bad) {
    const uint64_t i0 = w * 27;
    const int m = (n_len - i0) < 27 ? (int)(n_len - i0) : 27;
    uint32_t c[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) {
        uint32_t x = 0, pad = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * d + j;
            if (k < 27 && k < m) x |= (uint32_t)n[i0 + k] << (8 * j);
            else pad |= 0x41u << (8 * j);
        }
        bad += __builtin_popcount(invalid_mask<true>(x | pad));
        c[d] = lut ? code5_strict(x) : code5_fast(x);
    }
    return pack27(c);
}

// min(27, len - 27w) letters of packed word w with byte stores; bits beyond `len` are ignored
__device__ __forceinline__ void decode2_word_bytes(const uint64_t* __restrict__ bits, uint64_t len, uint8_t* __restrict__ out, uint64_t w) {
    const uint64_t i0 = w * 27;
    const uint64_t word = bits[w];
    const int m = (len - i0) < 27 ? (int)(len - i0) : 27;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        uint32_t l = letters5(digits3((uint32_t)(word >> (7 * t)) & 0x7Fu));
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (3 * t + j < m) out[i0 + 3 * t + j] = (uint8_t)(l >> (8 * j));
    }
}

// ---------------------------------------------------------------------------
// Generic kernels: one thread per word, byte accesses, any alignment/length.  Inputs below one tile, small ragged
// inputs, and the edges of the multi-wave variants; the default (one-wave) tile kernels carry their edges themselves.
// ---------------------------------------------------------------------------
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits2_generic(const uint8_t* __restrict__ n, uint64_t n_len,
                                                             uint64_t* __restrict__ out, uint64_t first_word,
                                                             uint64_t n_words, uint64_t lut_from) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock)
        out[w] = encode2_word_bytes(n, n_len, w, STRICT || w >= lut_from);
}

template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits2_generic_checked(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t* __restrict__ out, uint64_t first_word,
                                                                     uint64_t n_words, uint64_t lut_from, unsigned long long* __restrict__ bad_out, uint32_t slot_mask) {
    bad_out += blockIdx.x & slot_mask;
    uint32_t bad = 0;
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock)
        out[w] = encode2_word_bytes_checked(n, n_len, w, STRICT || w >= lut_from, bad);
    wave_add_invalid(bad, bad_out);
}

__global__ __launch_bounds__(kBlock) void bits_to_n2_generic(const uint64_t* __restrict__ bits, uint64_t len,
                                                             uint8_t* __restrict__ out, uint64_t first_word,
                                                             uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock)
        decode2_word_bytes(bits, len, out, w);
}

// Edges in the tile kernel's own launch (codec2_kernels.hpp, EDGES): words [0, head_words) and [tail_first, words) of a
// call, shared by the last `groups` workgroups of the launch behind their tile's stores.  One-wave workgroups only
// (grid == tiles); the multi-wave variants keep their separate generic launches.
struct Encode2Edges {
    const uint8_t* n;
    uint64_t* out;
    uint64_t n_len, head_words, tail_first, words, lut_from;
    uint32_t groups;
};
struct Decode2Edges {
    const uint64_t* bits;
    uint8_t* out;
    uint64_t len, head_words, tail_first, words;
    uint32_t groups;
};
template <bool STRICT>
__device__ __forceinline__ void encode2_edges(const Encode2Edges& e, uint64_t idx, uint64_t stride) {
    const uint64_t items = e.head_words + (e.words - e.tail_first);
    for (uint64_t i = idx; i < items; i += stride) {
        const uint64_t w = i < e.head_words ? i : e.tail_first + (i - e.head_words);
        e.out[w] = encode2_word_bytes(e.n, e.n_len, w, STRICT || w >= e.lut_from);
    }
}
template <bool STRICT>
__device__ __forceinline__ uint32_t encode2_edges_checked(const Encode2Edges& e, uint64_t idx, uint64_t stride) {
    const uint64_t items = e.head_words + (e.words - e.tail_first);
    uint32_t bad = 0;
    for (uint64_t i = idx; i < items; i += stride) {
        const uint64_t w = i < e.head_words ? i : e.tail_first + (i - e.head_words);
        e.out[w] = encode2_word_bytes_checked(e.n, e.n_len, w, STRICT || w >= e.lut_from, bad);
    }
    return bad;
}
__device__ __forceinline__ void decode2_edges(const Decode2Edges& e, uint64_t idx, uint64_t stride) {
    const uint64_t items = e.head_words + (e.words - e.tail_first);
    for (uint64_t i = idx; i < items; i += stride) decode2_word_bytes(e.bits, e.len, e.out, i < e.head_words ? i : e.tail_first + (i - e.head_words));
}

// ---------------------------------------------------------------------------
// Wave-tiled kernels.  One wave owns WPL*64 consecutive words = WPL*1728 B of ASCII
// (WPL words per lane).  WPL is even so that every wave tile is a whole number of
// 128-B cache lines on BOTH sides (1728 B = 13.5 lines: with WPL = 1 neighbouring
// waves split a line, which costs ~30 % on this chip -- same effect as a misaligned
// buffer).  A workgroup of WAVES waves handles WAVES consecutive wave tiles; slabs are
// per wave, so the only synchronisation is the wave-level LDS fence.
// ---------------------------------------------------------------------------
constexpr int kWaveWords5 = 64;                 // words per wave per round
constexpr int kWaveBytes5 = 27 * kWaveWords5;   // 1728 B of ASCII per round
constexpr int kWaveVecs5 = kWaveBytes5 / 16;    // 108
constexpr int kWaveDwords5 = kWaveBytes5 / 4;   // 432
static_assert(kWaveBytes5 % 16 == 0, "a round must be whole 16-B vectors");

// The 27 bytes at byte position `byte0` of a wave's LDS slab -> one packed word: 8 dword reads
// cover them, a funnel shift removes the byte phase, v_perm_b32 maps 4 bytes at a time to codes.
// Reads dwords byte0/4 .. byte0/4 + 7 (the slabs are padded for the last lane).
template <bool STRICT>
__device__ __forceinline__ uint64_t word_from_slab(const uint32_t* my, uint32_t byte0) {
    const uint32_t q = byte0 >> 2, s8 = (byte0 & 3u) * 8u;
    uint32_t raw[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) raw[d] = my[q + d];
    uint32_t x[7], c[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) x[d] = __builtin_amdgcn_alignbit(raw[d + 1], raw[d], s8);  // funnel shift right by the byte phase
    if constexpr (STRICT) {
#pragma unroll
        for (int d = 0; d < 7; ++d) c[d] = code5_strict(x[d]);
    } else {
#pragma unroll
        for (int d = 0; d < 7; ++d) c[d] = code5_table(x[d]);
        // bytes >= 0x80 encode as 0 (n_to_bits2.rs:151): checked once per wave, fixed up only if seen
        const uint32_t any = (x[0] | x[1] | x[2] | x[3] | x[4] | x[5] | x[6]) & 0x80808080u;
        if (__builtin_amdgcn_ballot_w64(any != 0) != 0) {
#pragma unroll
            for (int d = 0; d < 7; ++d) c[d] = code5_clear_high(x[d], c[d]);
        }
    }
    return pack27(c);  // byte 27 (x[6]'s top byte, the next word's first letter) has weight 0
}

// Encode: WPL*108 coalesced 16-B loads into the wave's LDS slab; then, per round j, lane l
// reads the 8 dwords that cover the 27 bytes of word j*64+l, funnel-shifts them into place,
// maps 4 bytes at a time to codes with v_perm_b32, and stores one u64 (8 B per lane,
// 512 B per wave-instruction).
template <int WAVES, int WPL, int LAUX, int SAUX, bool STRICT, int C, bool CHECK>
__device__ __forceinline__ void n_to_bits2_wave_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_wave_tiles, uint32_t xs,
                                                     const Encode2Edges& e, unsigned long long* __restrict__ bad_out) {
    constexpr int TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t slab[WAVES][kWaveDwords5 * WPL + 4];
    // readfirstlane makes the wave index provably wave-uniform: without it hipcc wraps every buffer
    // access whose descriptor depends on it in a waterfall loop (v_readfirstlane / s_and_saveexec)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // C > 1 (single-wave workgroups only): XCD-pair tile map, see tile_of_block
    const uint64_t t = C > 1 ? tile_of_block<C>(blockIdx.x, (uint32_t)n_wave_tiles, xs) : blockIdx.x * (uint64_t)WAVES + wave;
    if (t >= n_wave_tiles) return;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, TILE_BYTES);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    uint32_t* my = slab[wave];
    constexpr int NLD = (TILE_VECS + 63) / 64;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        v[i] = u32x4{0, 0, 0, 0};
        if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64))
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
    wave_lds_fence();
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        // byte 27*lane + 1728*j of the slab; 1728 is dword aligned: same phase every round
        const uint64_t word = word_from_slab<STRICT>(my, 27u * lane + (uint32_t)kWaveBytes5 * j);
        const vu2 w2 = {(uint32_t)word, (uint32_t)(word >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
    }
    if constexpr (CHECK) {
        // counted on the 16-B vectors as they were loaded -- an exact partition of the tile's bytes, no 27-byte bookkeeping;
        // the alphabet is the 5-letter codec's: ACGTUNacgtun (n_to_bits2.rs:8-23)
        static_assert(WAVES == 1, "the checked twin exists for the one-wave shape only");
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const bool mine = (i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64);
            sus += mine ? suspect16<true>(v[i], 0u) : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                const bool mine = (i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64);
                bad += mine ? invalid16<true>(v[i]) : 0u;
            }
        }
        if (blockIdx.x + e.groups >= n_wave_tiles)
            bad += encode2_edges_checked<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_wave_tiles) * 64 + lane, (uint64_t)e.groups * 64);
        wave_add_invalid(bad, bad_out);
    } else if constexpr (WAVES == 1) {  // grid == tiles: the launch's last e.groups workgroups share the edge words
        if (blockIdx.x + e.groups >= n_wave_tiles)
            encode2_edges<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_wave_tiles) * 64 + lane, (uint64_t)e.groups * 64);
    }
}
template <int WAVES, int WPL, int LAUX, int SAUX, bool STRICT, int C = 1>
__global__ __launch_bounds__(WAVES * 64) void n_to_bits2_wave(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               uint64_t n_wave_tiles, uint32_t xs, Encode2Edges e) {
    n_to_bits2_wave_body<WAVES, WPL, LAUX, SAUX, STRICT, C, false>(in, out, n_wave_tiles, xs, e, nullptr);
}
// CHECKED (round 6; cnt_n_to_bits2_checked_dev): *bad += the tile's bytes outside ACGTUNacgtun
template <int WPL, int LAUX, int SAUX, bool STRICT, int C = 1>
__global__ __launch_bounds__(64) void n_to_bits2_wave_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_wave_tiles, uint32_t xs,
                                                              Encode2Edges e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    n_to_bits2_wave_body<1, WPL, LAUX, SAUX, STRICT, C, true>(in, out, n_wave_tiles, xs, e, bad + (blockIdx.x & slot_mask));
}

// WINDOW: the default shape (one wave, 2 words per lane, 3456 B in, 1 KiB out) for an input at
// ANY byte address -- the 5-letter counterpart of n_to_bits_window.  `in` is the caller's
// pointer rounded down to 128 B and `phase` (1..127) the bytes dropped; the wave stages the
// aligned window that covers its tile (216 + 8 vectors) in its slab, and since every lane
// already picks its 27 bytes out of the slab at an arbitrary byte position, the phase is just an
// offset into it.  Reads up to 127 B before and 128 B behind the tile (launcher's business).
constexpr int kWindowSlabDwords5 = 4 * 64 * 4 + 4;
template <int LAUX, int SAUX, bool STRICT, int C, bool CHECK>
__device__ __forceinline__ void n_to_bits2_window_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_wave_tiles, uint32_t phase,
                                                       uint32_t xs, const Encode2Edges& e, unsigned long long* __restrict__ bad_out) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_WORDS = kWaveWords5 * WPL, WIN_VECS = kWaveVecs5 * WPL + 8;
    constexpr int NLD = (WIN_VECS + 63) / 64;
    // Branch-free on purpose: the fourth 16-B access -- 32 lanes' worth -- is issued by ALL lanes against a descriptor that
    // ends with the window (lanes 32..63 fall outside: zeros, no memory access), and the slab holds four whole wave rows so
    // that the LDS side needs no lane mask either.  With lane-masked accesses the compiler split the wave into two exec
    // branches and issued the fourth load only after lanes 32..63 had WAITED for the first three (two dependent trips to
    // memory per tile: 4.0 ms against the aligned kernel's 3.4 at 2^34 nt, profiles/r04_align_two_pass.jsonl).
    __shared__ __attribute__((aligned(16))) uint32_t my[kWindowSlabDwords5];
    static_assert(NLD * 64 * 4 + 4 == kWindowSlabDwords5, "four wave rows + the read-ahead of the last lane");
    const uint32_t lane = threadIdx.x;
    const uint64_t t = tile_of_block<C>(blockIdx.x, (uint32_t)n_wave_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, WIN_VECS * 16);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
#pragma unroll
    for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
    wave_lds_fence();
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        const uint64_t word = word_from_slab<STRICT>(my, phase + 27u * lane + (uint32_t)kWaveBytes5 * j);  // <= 3556
        const vu2 w2 = {(uint32_t)word, (uint32_t)(word >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
    }
    if constexpr (CHECK) {
        // the tile owns window bytes [phase, TILE_BYTES + phase): the first row from `phase` on, the last row up to it
        // (lanes 32..63 of that row hold the descriptor's zeros and lie behind the range anyway)
        static_assert(NLD == 4 && TILE_BYTES == 3456, "the ranges below are spelled for four rows over a 3456-byte tile");
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int i = 0; i < NLD - 1; ++i) sus = suspect16<true>(v[i], sus);
        sus += lane < (uint32_t)(WIN_VECS - (NLD - 1) * 64) ? suspect16<true>(v[NLD - 1], 0u) : 0u;
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {
            bad = invalid16_range<true>(v[0], (int)phase - 16 * (int)lane, 16) +
                  invalid16_range<true>(v[NLD - 1], 0, TILE_BYTES + (int)phase - (NLD - 1) * 1024 - 16 * (int)lane);
#pragma unroll
            for (int i = 1; i < NLD - 1; ++i) bad += invalid16<true>(v[i]);
        }
        if (blockIdx.x + e.groups >= n_wave_tiles)
            bad += encode2_edges_checked<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_wave_tiles) * 64 + lane, (uint64_t)e.groups * 64);
        wave_add_invalid(bad, bad_out);
    } else {
        if (blockIdx.x + e.groups >= n_wave_tiles)
            encode2_edges<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_wave_tiles) * 64 + lane, (uint64_t)e.groups * 64);
    }
}
template <int LAUX, int SAUX, bool STRICT, int C>
__global__ __launch_bounds__(64) void n_to_bits2_window(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        uint64_t n_wave_tiles, uint32_t phase, uint32_t xs, Encode2Edges e) {
    n_to_bits2_window_body<LAUX, SAUX, STRICT, C, false>(in, out, n_wave_tiles, phase, xs, e, nullptr);
}
template <int LAUX, int SAUX, bool STRICT, int C>
__global__ __launch_bounds__(64) void n_to_bits2_window_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_wave_tiles, uint32_t phase,
                                                                uint32_t xs, Encode2Edges e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    n_to_bits2_window_body<LAUX, SAUX, STRICT, C, true>(in, out, n_wave_tiles, phase, xs, e, bad + (blockIdx.x & slot_mask));
}

// Decode: per round j, lane l loads word j*64+l (8 B, 512 B per wave-instruction), expands
// it to 27 letters held as 7 dwords, shifts them to the byte phase of position 27*l,
// completes the dword it shares with lane l-1 by one wave shuffle, and writes 6-7 ALIGNED
// dwords to the wave's LDS slab (every slab dword is written by exactly one lane -- no byte
// stores; rounds are independent because 1728 B is dword aligned).  The tile then leaves
// with WPL*108 coalesced 16-B stores.
template <int WAVES, int WPL, int LAUX, int SAUX, int C = 1>
__global__ __launch_bounds__(WAVES * 64) void bits_to_n2_wave(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               uint64_t n_wave_tiles, uint32_t xs, Decode2Edges e) {
    constexpr int TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t slab[WAVES][kWaveDwords5 * WPL + 4];
    // readfirstlane makes the wave index provably wave-uniform: without it hipcc wraps every buffer
    // access whose descriptor depends on it in a waterfall loop (v_readfirstlane / s_and_saveexec)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // C > 1 (single-wave workgroups only): XCD-pair tile map, see tile_of_block
    const uint64_t t = C > 1 ? tile_of_block<C>(blockIdx.x, (uint32_t)n_wave_tiles, xs) : blockIdx.x * (uint64_t)WAVES + wave;
    if (t >= n_wave_tiles) return;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_BYTES, TILE_BYTES);
    uint32_t* my = slab[wave];
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    vu2 w2[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) w2[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
    // per-lane constants: 1728 B per round is dword aligned, so every round has the same phase.
    // The lane's 27 bytes start at byte 27*lane = dword q0, byte phase ph; shifting the string up by
    // ph bytes is one v_perm_b32 per dword with selector bytes (4-ph, 5-ph, 6-ph, 7-ph) over
    // {b[k], b[k-1]}.  The lane owns the complete dwords q0 .. ((27*(lane+1))>>2) - 1 (6 or 7 of them);
    // the partial one behind them belongs to lane+1, which receives it through the shuffle.
    const uint32_t byte0 = 27u * lane, q0 = byte0 >> 2, ph = byte0 & 3u;
    const uint32_t sel = 0x07060504u - 0x01010101u * ph;
    const uint32_t cnt = ((byte0 + 27u) >> 2) - q0;  // 6 or 7
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        uint32_t b[7];
        decode27(w2[j][0], w2[j][1], b);
        uint32_t W[8];
        W[0] = __builtin_amdgcn_perm(b[0], 0u, sel);
#pragma unroll
        for (int k = 1; k < 7; ++k) W[k] = __builtin_amdgcn_perm(b[k], b[k - 1], sel);
        W[7] = __builtin_amdgcn_perm(0u, b[6], sel);
        const uint32_t tail = cnt == 6 ? W[6] : W[7];
        const uint32_t prev_tail = __shfl_up(tail, 1, 64);
        if (ph != 0) W[0] |= prev_tail;  // lane 0 has phase 0
        uint32_t* dst = my + kWaveDwords5 * j + q0;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k] = W[k];
        if (cnt == 7) dst[6] = W[6];
    }
    wave_lds_fence();
    constexpr int NST = (TILE_VECS + 63) / 64;
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(my + (i * 64 + lane) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
        }
    if constexpr (WAVES == 1) {
        if (blockIdx.x + e.groups >= n_wave_tiles)
            decode2_edges(e, (uint64_t)(blockIdx.x + e.groups - n_wave_tiles) * 64 + lane, (uint64_t)e.groups * 64);
    }
}

// ---------------------------------------------------------------------------
// PAGE tiles for the decoder (round 4).  The word tiles above give every wave 3456 B = 27 lines of the ASCII stream, so
// each 4-KiB page of the stream that carries 77 % of the bytes is written by two waves; an arithmetic-free lab (profiles/r04_codec5_page_tiles_probe.jsonl) prices
// the two address patterns without arithmetic: one wave per whole page of the WRITE stream (four full 1-KiB stores -- the
// 2-bit decoder's shape) and a ragged 1213.6-B piece of the packed stream runs 3 % faster than the word tiles.  Here
// tile t owns letters [4096 t, 4096 t + 4096) of the launch: its first word is w0 = 4096 t / 27, the page starts r =
// 4096 t - 27 w0 letters into it, and 153 words (three rounds of 64 lanes, the third with 25) always cover r + 4096 <=
// 4122 letters.  Word idx is expanded exactly as in bits_to_n2_wave but lands at slab byte 27 idx + 32 - r, which puts the
// page's first letter on slab byte 32 for every r: the page leaves with four aligned 16-B reads per lane.  Rounds keep
// one byte phase (1728 B per round is dword aligned); what lane 0 of round j shares with lane 63 of round j - 1 comes
// through one v_readlane.  Words past the end of the packed array read as 0 through the descriptor; their letters lie
// behind the page.
// ---------------------------------------------------------------------------
constexpr int kPageNt5 = 4096;
// PAGES consecutive pages per wave: 153 words in 3 rounds of 64 lanes (the third keeps 25 busy) or 305 in 5 (49 in the fifth)
template <int PAGES> struct PageTile5 {
    static_assert(PAGES == 1 || PAGES == 2, "one or two pages per wave");
    static constexpr int kNt = PAGES * kPageNt5, kWords = (26 + kNt + 26) / 27, kRounds = (kWords + 63) / 64;
    static constexpr int kSlabDwords = (32 + 27 * 64 * kRounds + 3) / 4 + 4;
};
struct Decode2PageEdges {
    const uint64_t* bits;
    uint8_t* out;
    uint64_t len, head_nt, tail_from;  // letters [0, head_nt) and [tail_from, len) belong to the edge items
    uint32_t groups;
};
// letters [lo, hi) n word w's 27 with byte stores
__device__ __forceinline__ void decode2_word_bytes_range(const uint64_t* __restrict__ bits, uint64_t lo, uint64_t hi, uint8_t* __restrict__ out, uint64_t w) {
    const uint64_t i0 = w * 27;
    const uint64_t word = bits[w];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        uint32_t l = letters5(digits3((uint32_t)(word >> (7 * t)) & 0x7Fu));
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (i0 + 3 * t + j >= lo && i0 + 3 * t + j < hi) out[i0 + 3 * t + j] = (uint8_t)(l >> (8 * j));
    }
}
__device__ __forceinline__ void decode2_page_edges(const Decode2PageEdges& e, uint64_t idx, uint64_t stride) {
    const uint64_t head_words = (e.head_nt + 26) / 27, tail_w0 = e.tail_from / 27, tail_words = e.tail_from < e.len ? (e.len + 26) / 27 - tail_w0 : 0;
    for (uint64_t i = idx; i < head_words + tail_words; i += stride) {
        if (i < head_words) decode2_word_bytes_range(e.bits, 0, e.head_nt, e.out, i);
        else decode2_word_bytes_range(e.bits, e.tail_from, e.len, e.out, tail_w0 + (i - head_words));
    }
}
// `bits` = the call's packed array (8-B aligned), `words` its length; `out` + nt0 is 128-B aligned, nt0 = the first letter
// of this launch's first page
template <int C, int LAUX, int SAUX, int PAGES = 1>
__global__ __launch_bounds__(64) void bits_to_n2_page(const uint64_t* __restrict__ bits, uint64_t words, uint8_t* __restrict__ out, uint64_t nt0,
                                                       uint32_t n_tiles, uint32_t xs, Decode2PageEdges e) {
    using T = PageTile5<PAGES>;
    __shared__ __attribute__((aligned(16))) uint32_t my[T::kSlabDwords];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const uint64_t l0 = nt0 + t * T::kNt, w0 = l0 / 27;
    const uint32_t r = (uint32_t)(l0 - w0 * 27);
    const uint64_t left = words - w0;  // >= 1: the page lies inside the decoded length
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(bits + w0, (uint32_t)(left < (uint64_t)T::kWords ? left : (uint64_t)T::kWords) * 8);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + l0, T::kNt);
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    vu2 w2[T::kRounds];
#pragma unroll
    for (int j = 0; j < T::kRounds; ++j) w2[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
    const uint32_t byte0 = 27u * lane + 32u - r, q0 = byte0 >> 2, ph = byte0 & 3u;
    const uint32_t sel = 0x07060504u - 0x01010101u * ph;
    const uint32_t cnt = ((byte0 + 27u) >> 2) - q0;  // 6 or 7
    uint32_t carry = 0;  // round 0, lane 0: the bytes in front of word w0 lie in front of the page
#pragma unroll
    for (int j = 0; j < T::kRounds; ++j) {
        uint32_t b[7];
        decode27(w2[j][0], w2[j][1], b);
        uint32_t W[8];
        W[0] = __builtin_amdgcn_perm(b[0], 0u, sel);
#pragma unroll
        for (int k = 1; k < 7; ++k) W[k] = __builtin_amdgcn_perm(b[k], b[k - 1], sel);
        W[7] = __builtin_amdgcn_perm(0u, b[6], sel);
        const uint32_t tail = cnt == 6 ? W[6] : W[7];
        uint32_t prev_tail = __shfl_up(tail, 1, 64);
        if (lane == 0) prev_tail = carry;
        carry = __builtin_amdgcn_readlane(tail, 63);
        if (ph != 0) W[0] |= prev_tail;
        uint32_t* dst = my + kWaveDwords5 * j + q0;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k] = W[k];
        if (cnt == 7) dst[6] = W[6];
    }
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 4 * PAGES; ++i) {
        const u32x4 o = *reinterpret_cast<const u32x4*>(my + 8 + (i * 64 + lane) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
    }
    if (blockIdx.x + e.groups >= n_tiles)
        decode2_page_edges(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * 64 + lane, (uint64_t)e.groups * 64);
}

// ---------------------------------------------------------------------------
// PIPELINED wave tiles (round 4).  The L2 <-> fabric counters (profiles/r04_bound_counters_codec5.json) show both
// directions IN-FLIGHT starved, not pushed back by the memory side: the encoder keeps 36.6k reads outstanding at the
// 2-bit encoder's latency (0.93 us) where that one keeps 39.2k, the decoder 12.8k writes where the 2-bit decoder keeps
// 15.4k, with write-credit stalls at 0.6 % of L2-busy against 5.6 % -- a wave that is staging through LDS and doing
// base-5 arithmetic has nothing in flight, and that phase is long here.  So one wave takes K CONSECUTIVE tiles and
// issues tile i+1's global loads BEFORE it touches tile i's data: the arithmetic of a tile runs under the next
// tile's read latency.  One slab per wave is enough (a wave's LDS instructions execute in issue order: tile i+1's
// ds_writes cannot pass tile i's ds_reads; the fences only pin the compiler).  Loads return in order, so waiting for
// tile i's vectors leaves tile i+1's outstanding (vmcnt counts down in issue order; tile i's stores are issued after
// tile i+1's loads and never gate them).
// ---------------------------------------------------------------------------
// Branch-free on purpose: the grid covers whole groups of K tiles only (the launcher hands the < K leftover tiles to the
// edge words), and the fourth 16-B access of a 3456-B tile -- 24 lanes' worth -- is issued by ALL lanes against a
// descriptor that ends with the tile: lanes 24..63 fall outside, read zeros / drop their stores without touching memory,
// and the slab is a full 4 KiB so the LDS side needs no lane mask either.  (With the lane-masked accesses of
// n_to_bits2_wave the compiler's s_waitcnt insertion lost track across the exec branches and waited for tile i+1's loads
// before staging tile i -- vmcnt(2..0) where vmcnt(7..4) was meant -- which is the whole point gone.)
constexpr int kPipeSlabDwords5 = 1024 + 4;  // 4 wave-wide 16-B rows + the 8-dword read-ahead of the last lane
template <int K, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(64) void n_to_bits2_pipe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, Encode2Edges e) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_WORDS = kWaveWords5 * WPL, NLD = 4;
    static_assert((kWaveVecs5 * WPL + 63) / 64 == NLD, "four wave-wide rows per tile");
    __shared__ __attribute__((aligned(16))) uint32_t my[kPipeSlabDwords5];
    const uint32_t lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * K;
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    u32x4 v[K][NLD];
    auto issue = [&](int k) {
        const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (t0 + k) * TILE_BYTES, TILE_BYTES);
#pragma unroll
        for (int i = 0; i < NLD; ++i) v[k][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    };
    issue(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k + 1 < K) issue(k + 1);
#pragma unroll
        for (int i = 0; i < NLD; ++i) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[k][i];
        wave_lds_fence();
        const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + (t0 + k) * (TILE_WORDS * 8), TILE_WORDS * 8);
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const uint64_t word = word_from_slab<STRICT>(my, 27u * lane + (uint32_t)kWaveBytes5 * j);
            const vu2 w2 = {(uint32_t)word, (uint32_t)(word >> 32)};
            __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
        }
        wave_lds_fence();
    }
    if (blockIdx.x + e.groups >= gridDim.x)
        encode2_edges<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - gridDim.x) * 64 + lane, (uint64_t)e.groups * 64);
}

template <int K, int LAUX, int SAUX>
__global__ __launch_bounds__(64) void bits_to_n2_pipe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, Decode2Edges e) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_WORDS = kWaveWords5 * WPL, NST = 4;
    __shared__ __attribute__((aligned(16))) uint32_t my[kPipeSlabDwords5];
    const uint32_t lane = threadIdx.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * K;
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    vu2 w[K][WPL];
    auto issue = [&](int k) {
        const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (t0 + k) * (TILE_WORDS * 8), TILE_WORDS * 8);
#pragma unroll
        for (int j = 0; j < WPL; ++j) w[k][j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
    };
    const uint32_t byte0 = 27u * lane, q0 = byte0 >> 2, ph = byte0 & 3u;
    const uint32_t sel = 0x07060504u - 0x01010101u * ph;
    const uint32_t cnt = ((byte0 + 27u) >> 2) - q0;  // 6 or 7
    issue(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (k + 1 < K) issue(k + 1);
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            uint32_t b[7];
            decode27(w[k][j][0], w[k][j][1], b);
            uint32_t W[8];
            W[0] = __builtin_amdgcn_perm(b[0], 0u, sel);
#pragma unroll
            for (int q = 1; q < 7; ++q) W[q] = __builtin_amdgcn_perm(b[q], b[q - 1], sel);
            W[7] = __builtin_amdgcn_perm(0u, b[6], sel);
            const uint32_t tail = cnt == 6 ? W[6] : W[7];
            const uint32_t prev_tail = __shfl_up(tail, 1, 64);
            if (ph != 0) W[0] |= prev_tail;
            uint32_t* dst = my + kWaveDwords5 * j + q0;
#pragma unroll
            for (int q = 0; q < 6; ++q) dst[q] = W[q];
            if (cnt == 7) dst[6] = W[6];
        }
        wave_lds_fence();
        const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + (t0 + k) * TILE_BYTES, TILE_BYTES);  // lanes 24..63 of the 4th store fall outside: dropped
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(my + (i * 64 + lane) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
        }
        wave_lds_fence();
    }
    if (blockIdx.x + e.groups >= gridDim.x)
        decode2_edges(e, (uint64_t)(blockIdx.x + e.groups - gridDim.x) * 64 + lane, (uint64_t)e.groups * 64);
}

}  // namespace cnt
