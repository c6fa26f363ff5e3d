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

template <bool STRICT>
__device__ __forceinline__ uint32_t code5(uint32_t x) {
    if constexpr (STRICT) return code5_strict(x);
    else return code5_fast(x);
}

// 27 codes (one per byte, 7 dwords, bytes past the end = 0) -> one packed word
__device__ __forceinline__ uint64_t pack27(const uint32_t (&c)[7]) {
    uint64_t acc = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = 3 * t + j;
            const uint32_t code = (c[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            v += code * (j == 0 ? 1u : j == 1 ? 5u : 25u);
        }
        acc |= (uint64_t)v << (7 * t);
    }
    return acc;
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

// ---------------------------------------------------------------------------
// Generic kernels: one thread per word, byte accesses, any alignment/length.
// ---------------------------------------------------------------------------
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits2_generic(const uint8_t* __restrict__ n, uint64_t n_len,
                                                             uint64_t* __restrict__ out, uint64_t first_word,
                                                             uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock) {
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
            c[d] = code5<STRICT>(x);  // unloaded bytes are 0 -> code 0 (missing digits = 0, n_to_bits2.rs:58-70)
        }
        out[w] = pack27(c);
    }
}

__global__ __launch_bounds__(kBlock) void bits_to_n2_generic(const uint64_t* __restrict__ bits, uint64_t len,
                                                             uint8_t* __restrict__ out, uint64_t first_word,
                                                             uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock) {
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
    uint32_t c[7];
#pragma unroll
    for (int d = 0; d < 7; ++d) {
        uint32_t x = (uint32_t)((((uint64_t)raw[d + 1] << 32) | raw[d]) >> s8);  // funnel shift right by the byte phase
        if (d == 6) x &= 0x00FFFFFFu;                                             // bytes 24..26 only
        c[d] = code5<STRICT>(x);
    }
    return pack27(c);
}

// Encode: WPL*108 coalesced 16-B loads into the wave's LDS slab; then, per round j, lane l
// reads the 8 dwords that cover the 27 bytes of word j*64+l, funnel-shifts them into place,
// maps 4 bytes at a time to codes with v_perm_b32, and stores one u64 (8 B per lane,
// 512 B per wave-instruction).
template <int WAVES, int WPL, int LAUX, int SAUX, bool STRICT, int C = 1>
__global__ __launch_bounds__(WAVES * 64) void n_to_bits2_wave(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               uint64_t n_wave_tiles) {
    constexpr int TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t slab[WAVES][kWaveDwords5 * WPL + 4];
    // readfirstlane makes the wave index provably wave-uniform: without it hipcc wraps every buffer
    // access whose descriptor depends on it in a waterfall loop (v_readfirstlane / s_and_saveexec)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // C > 1 (single-wave workgroups only): XCD-pair tile map, see tile_of_block
    const uint64_t t = C > 1 ? tile_of_block<C>(blockIdx.x, n_wave_tiles) : blockIdx.x * (uint64_t)WAVES + wave;
    if (t >= n_wave_tiles) return;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, TILE_BYTES);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    uint32_t* my = slab[wave];
    constexpr int NLD = (TILE_VECS + 63) / 64;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        v[i] = u32x4{0, 0, 0, 0};
        if ((i + 1) * 64 <= TILE_VECS || lane < TILE_VECS - i * 64)
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        if ((i + 1) * 64 <= TILE_VECS || lane < TILE_VECS - i * 64) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
    wave_lds_fence();
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        // byte 27*lane + 1728*j of the slab; 1728 is dword aligned: same phase every round
        const uint64_t word = word_from_slab<STRICT>(my, 27u * lane + (uint32_t)kWaveBytes5 * j);
        const vu2 w2 = {(uint32_t)word, (uint32_t)(word >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
    }
}

// WINDOW: the default shape (one wave, 2 words per lane, 3456 B in, 1 KiB out) for an input at
// ANY byte address -- the 5-letter counterpart of n_to_bits_window.  `in` is the caller's
// pointer rounded down to 128 B and `phase` (1..127) the bytes dropped; the wave stages the
// aligned window that covers its tile (216 + 8 vectors) in its slab, and since every lane
// already picks its 27 bytes out of the slab at an arbitrary byte position, the phase is just an
// offset into it.  Reads up to 127 B before and 128 B behind the tile (launcher's business).
template <int LAUX, int SAUX, bool STRICT, int C>
__global__ __launch_bounds__(64) void n_to_bits2_window(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                        uint64_t n_wave_tiles, uint32_t phase) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_WORDS = kWaveWords5 * WPL, WIN_VECS = kWaveVecs5 * WPL + 8;
    __shared__ __attribute__((aligned(16))) uint32_t my[WIN_VECS * 4 + 4];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_wave_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, WIN_VECS * 16);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    constexpr int NLD = (WIN_VECS + 63) / 64;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        v[i] = u32x4{0, 0, 0, 0};
        if ((i + 1) * 64 <= WIN_VECS || lane < WIN_VECS - i * 64)
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i)
        if ((i + 1) * 64 <= WIN_VECS || lane < WIN_VECS - i * 64) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
    wave_lds_fence();
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        const uint64_t word = word_from_slab<STRICT>(my, phase + 27u * lane + (uint32_t)kWaveBytes5 * j);  // <= 3556
        const vu2 w2 = {(uint32_t)word, (uint32_t)(word >> 32)};
        __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
    }
}

// Decode: per round j, lane l loads word j*64+l (8 B, 512 B per wave-instruction), expands
// it to 27 letters held as 7 dwords, shifts them to the byte phase of position 27*l,
// completes the dword it shares with lane l-1 by one wave shuffle, and writes 6-7 ALIGNED
// dwords to the wave's LDS slab (every slab dword is written by exactly one lane -- no byte
// stores; rounds are independent because 1728 B is dword aligned).  The tile then leaves
// with WPL*108 coalesced 16-B stores.
template <int WAVES, int WPL, int LAUX, int SAUX, int C = 1>
__global__ __launch_bounds__(WAVES * 64) void bits_to_n2_wave(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               uint64_t n_wave_tiles) {
    constexpr int TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t slab[WAVES][kWaveDwords5 * WPL + 4];
    // readfirstlane makes the wave index provably wave-uniform: without it hipcc wraps every buffer
    // access whose descriptor depends on it in a waterfall loop (v_readfirstlane / s_and_saveexec)
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // C > 1 (single-wave workgroups only): XCD-pair tile map, see tile_of_block
    const uint64_t t = C > 1 ? tile_of_block<C>(blockIdx.x, n_wave_tiles) : blockIdx.x * (uint64_t)WAVES + wave;
    if (t >= n_wave_tiles) return;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_BYTES, TILE_BYTES);
    uint32_t* my = slab[wave];
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    vu2 w2[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) w2[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
        const uint32_t lo = w2[j][0], hi = w2[j][1];
        uint32_t L[9];  // 3 letters per triplet in the low 24 bits
#define CNT_L(K) L[K] = letters5(digits3(field7<K>(lo, hi))) & 0x00FFFFFFu
        CNT_L(0); CNT_L(1); CNT_L(2); CNT_L(3); CNT_L(4); CNT_L(5); CNT_L(6); CNT_L(7); CNT_L(8);
#undef CNT_L
        // the lane's 27 bytes as 7 dwords (b[6] holds 3 bytes), plus a zero guard
        uint32_t b[8];
        b[0] = L[0] | (L[1] << 24);
        b[1] = (L[1] >> 8) | (L[2] << 16);
        b[2] = (L[2] >> 16) | (L[3] << 8);
        b[3] = L[4] | (L[5] << 24);
        b[4] = (L[5] >> 8) | (L[6] << 16);
        b[5] = (L[6] >> 16) | (L[7] << 8);
        b[6] = L[8];
        b[7] = 0;
        // window of 8 aligned dwords starting at dword (27*lane)>>2, bytes shifted left by the phase
        const uint32_t byte0 = 27u * lane, q0 = byte0 >> 2, s8 = (byte0 & 3u) * 8u;
        // W = the 27-byte string shifted left by the phase: one v_alignbit_b32 per dword ({b[k],
        // b[k-1]} >> (32 - s8)); a phase of 0 would need a shift of 32, which alignbit cannot do
        uint32_t W[8];
        const uint32_t sh = 32u - s8;
        W[0] = b[0] << s8;
#pragma unroll
        for (int k = 1; k < 8; ++k) W[k] = s8 ? __builtin_amdgcn_alignbit(b[k], b[k - 1], sh) : b[k];
        // complete dwords this lane owns: q0 .. ((27*(lane+1))>>2) - 1 (6 or 7 of them); the
        // partial one after them belongs to lane+1, which receives it through the shuffle
        const uint32_t cnt = ((byte0 + 27u) >> 2) - q0;  // 6 or 7
        const uint32_t tail = cnt == 6 ? W[6] : W[7];
        const uint32_t prev_tail = __shfl_up(tail, 1, 64);
        if (s8 != 0) W[0] |= prev_tail;  // lane 0 has s8 == 0
        uint32_t* dst = my + kWaveDwords5 * j + q0;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k] = W[k];
        if (cnt == 7) dst[6] = W[6];
    }
    wave_lds_fence();
    constexpr int NST = (TILE_VECS + 63) / 64;
#pragma unroll
    for (int i = 0; i < NST; ++i)
        if ((i + 1) * 64 <= TILE_VECS || lane < TILE_VECS - i * 64) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(my + (i * 64 + lane) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
        }
}

}  // namespace cnt
