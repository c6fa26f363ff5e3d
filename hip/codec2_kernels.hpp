// codec2_kernels.hpp -- gfx950 kernels for the 2-bit nucleotide codec.
//
// Replaces the loops of n_to_bits_{lut,pext,shift,movemask,mul} (reference
// src/n_to_bits.rs:34-259) and bits_to_n_{lut,shuffle,pdep,clmul} (:51-69,
// :265-404).  Not a translation of the AVX2 code: the four x86 encoders are four
// routes to one function -- out_byte[j] = c(4j) | c(4j+1)<<2 | c(4j+2)<<4 |
// c(4j+3)<<6 with c = (ascii>>1)&3 -- and because u64 words are little-endian on
// both machines the packed output is just that byte stream.  So the GPU kernels
// work on dwords: one input dword (4 nt) <-> one packed byte.
//
// Both directions are pure HBM streams (1.25 B/nt, zero reuse, ~5 VALU ops per
// dword against ~60 available per dword at full bandwidth), so the kernels are
// shaped by the memory system alone.  What the A/B labs on MI355X found
// (bench/tune_lab*.hip, logs under profiles/, summary in DESIGN.md):
//   * few bytes per wave and MANY small workgroups in dispatch order beat deep
//     unrolling or persistent grid-stride loops: the set of tiles in flight is
//     then a compact, advancing address window;
//   * 16 B per lane on the wide side, the narrow side simply 4 B per lane --
//     staging the narrow side through LDS to widen it to 16 B bought nothing;
//   * cache policy matters: streaming (nt) loads for encode, plain loads for decode,
//     write-through non-temporal (sc0|sc1|nt) stores for both (for encode sc1 alone
//     is as fast, but leaves the packed buffer in a state that slows the decode that
//     follows by 1.3 %);
//   * which XCD touches which 4 KiB (block b is observed to run on XCD b % X): DECODE is 1.5-3 % faster when each XCD
//     takes its READ stream in whole 4-KiB pieces -- four 4-KiB output tiles (= 4 KiB of packed words) per turn; other
//     group sizes lose, and what the maps that win have in common is that the XCDs read CONSECUTIVE pages at a time, each
//     XCD whole pages, always of the same residue class (bench/tune_lab10.hip; the reductions in packed_ops_kernels.hpp
//     get the same property from vec_offset).  ENCODE shipped with the analogous pair map (two 2-KiB ASCII tiles per
//     turn) through rounds 1-2; under the residency cap and the store policy adopted AFTER it, plain dispatch order is
//     0.2-0.9 % faster in encode -> decode steps on five boxes out of five (round 3, bench/xcd_shift_ab.py,
//     profiles/r03_ab_step_encode_plain_order.log), so encode's default has no map; the pair map is variant 17.  Class
//     affinity without the consecutive order loses (tried on the 5-letter encoder, profiles/HISTORY.md 5);
//   * capping residency at ~24 waves per CU (dummy LDS) is worth another 2-3 %.
// Global accesses go through raw buffer loads/stores: a wave-uniform descriptor
// per tile gives 32-bit lane offsets under a 64-bit tile base (2^36-nt buffers)
// and exposes the sc0 / nt / sc1 bits that plain C++ loads and stores cannot.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnt {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int vu4 __attribute__((__vector_size__(16)));  // the buffer builtins' 128-bit type

constexpr int kBlock = 256;  // generic / utility kernels: 4 waves, one per SIMD
constexpr int kWave = 64;

// cache-policy bits of the gfx950 buffer instructions (the builtin's `aux` operand)
constexpr int kSC0 = 1, kNT = 2, kSC1 = 16;

// ---------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------
// Wave-uniform buffer descriptor over [p, p+bytes): raw (untyped) buffer, no
// swizzle; out-of-range lanes read 0 / drop stores (not relied upon).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// Order a wave's LDS writes before its own later LDS reads of other lanes'
// data.  LDS instructions of one wave execute in issue order, so no s_barrier
// is needed -- only a compiler fence so the accesses are not reordered.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Block -> tile mapping.  Dispatch is observed (never relied on for correctness) to place block b on
// XCD b % X, X = the XCD count of the device (hipDeviceAttributeNumberOfXccs, a power of two: 8 on an
// MI355X in SPX mode, fewer in the partitioned modes; the launchers pass xs = log2 X, see chip_info()).
// With C > 1 each XCD turn covers C consecutive tiles, i.e. a C-tile contiguous piece per private L2.
// Bijective on [0, n_tiles): whole groups of X*C tiles are permuted, the ragged rest is left in place.
// All 32-bit: one launch never has more than 2^31-1 threads, i.e. < 2^26 tiles (max_tiles_per_launch).
template <int C>
__device__ __forceinline__ uint32_t tile_of_block(uint32_t b, uint32_t n_tiles, uint32_t xs) {
    if constexpr (C == 1) {
        return b;
    } else {
        static_assert(C == 2 || C == 4 || C == 8 || C == 16 || C == 32, "C is a power of two");
        constexpr uint32_t cs = C == 2 ? 1 : C == 4 ? 2 : C == 8 ? 3 : C == 16 ? 4 : 5;
        const uint32_t gs = xs + cs;  // log2 of the group size X*C
        const uint32_t g = b >> gs;
        if (((g + 1) << gs) > n_tiles) return b;
        // group base + (XCD slot x of the block) * C + (turn of that XCD inside the group); xs < gs, so b's low xs
        // bits are the slot directly
        const uint32_t x = b & ~(~0u << xs), turn = (b >> xs) & (uint32_t)(C - 1);
        return (g << gs) + (x << cs) + turn;
    }
}

// ---------------------------------------------------------------------------
// encode arithmetic: 4 ASCII bytes (one dword) -> one packed byte
// ---------------------------------------------------------------------------
// Bits 1..2 of each byte are the code (n_to_bits.rs:85 mask 0x06).  With
// y = x & 0x06060606 the four codes sit at bits {1,2},{9,10},{17,18},{25,26};
// OR-ing y, y<<6, y<<12, y<<18 lines them up at bits 19..26 (checked
// exhaustively in tests/test_bit_tricks.py).  The source spells it as two shift-ORs; because the
// shifted copies never overlap, OR == ADD and hipcc -O3 emits ONE v_mul_lo_u32 by 0x41041 (0x820820
// for the dword whose byte is wanted at bits 24..31) -- the reference's n_to_bits_mul identity
// (n_to_bits.rs:223-231) found by the compiler.  Quarter-rate, 8 per lane and tile, and free: forcing
// the two full-rate v_lshl_or_b32 with inline asm, or removing the arithmetic altogether, moves the
// kernel by < 0.3 % (bench/tune_lab12.hip, profiles/r02_tune_lab12_encode_arithmetic.log).
__device__ __forceinline__ uint32_t enc_gather(uint32_t y) {
    uint32_t u = (y << 6) | y;
    return (u << 12) | u;  // packed byte at bits 19..26
}

// CNT_STRICT_LUT: clear the code of every byte that is not one of ACGTUacgtu
// (n_to_bits.rs:8-21 gives those 0).  An 8-entry table indexed by the low three
// bits (A=1 C=3 T=4 U=5 G=7) via v_perm_b32 says which upper-case letter the
// byte would have to be; a SWAR zero-byte test compares all four at once.
__device__ __forceinline__ uint32_t strict_filter(uint32_t x) {
    // table byte k = the only valid (upper-case) letter whose low 3 bits are k, else 0xFF
    const uint32_t lut_lo = 0x43FF41FFu;  // k=0:FF 1:'A' 2:FF 3:'C'
    const uint32_t lut_hi = 0x47FF5554u;  // k=4:'T' 5:'U' 6:FF 7:'G'
    uint32_t expect = __builtin_amdgcn_perm(lut_hi, lut_lo, x & 0x07070707u);
    uint32_t z = (x & 0xDFDFDFDFu) ^ expect;                    // zero byte <=> valid letter (case folded)
    uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;  // 0x80 per NON-zero byte
    uint32_t kill = (nz >> 5) | (nz >> 6);                      // bits 2 and 1 of each invalid byte
    return x & ~kill;
}

template <bool STRICT>
__device__ __forceinline__ uint32_t enc4(uint32_t x) {  // returns packed byte at bits 19..26
    if constexpr (STRICT) x = strict_filter(x);
    return enc_gather(x & 0x06060606u);
}

// 16 ASCII bytes -> one packed dword
template <bool STRICT>
__device__ __forceinline__ uint32_t enc16(u32x4 q) {
    uint32_t b0 = __builtin_amdgcn_ubfe(enc4<STRICT>(q.x), 19, 8);
    uint32_t b1 = __builtin_amdgcn_ubfe(enc4<STRICT>(q.y), 19, 8);
    uint32_t b2 = __builtin_amdgcn_ubfe(enc4<STRICT>(q.z), 19, 8);
    uint32_t b3 = enc4<STRICT>(q.w) << 5;  // bits 24..31 (plus low garbage masked below)
    return b0 | (b1 << 8) | (b2 << 16) | (b3 & 0xFF000000u);
}

// ---------------------------------------------------------------------------
// validity while encoding (round 6): the *_checked entry points count the bytes outside the codec's alphabet in the
// SAME pass that packs them -- the reference's BYTE_LUT silently encodes every such byte as 0 (n_to_bits.rs:8-21,42) and
// points at a separate validity check (README.md:23); encode + cnt_validate_dev is 2.25 B/nt of HBM traffic for what one
// pass over the ASCII does at 1.25.
// ---------------------------------------------------------------------------
// 0x80 in every byte of x that is NOT a letter of the alphabet: ACGTUacgtu, with ALLOW_N also Nn (the 5-letter codec's,
// n_to_bits2.rs:8-23).  strict_filter's test: the low three bits say which upper-case letter the byte would have to be.
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t invalid_mask(uint32_t x) {
    const uint32_t expect = __builtin_amdgcn_perm(ALLOW_N ? 0x474E5554u : 0x47FF5554u, 0x43FF41FFu, x & 0x07070707u);
    const uint32_t z = (x & 0xDFDFDFDFu) ^ expect;                      // zero byte <=> valid letter (case folded)
    return (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
}
// The tiles' fast path does not count, it only finds out whether there is anything to count: `acc` grows by the byte-wise
// |folded - expected| of the four bytes (v_sad_u8: and, and, perm, sad = 4 VALU per dword against 8 for the exact mask +
// v_bcnt), so acc stays 0 exactly when every byte seen is a letter.  Clean data -- the case a validated encode is run for --
// then costs one wave-uniform branch behind the tile's stores; a wave that saw anything recounts its registers exactly.
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t suspect(uint32_t x, uint32_t acc) {
    const uint32_t expect = __builtin_amdgcn_perm(ALLOW_N ? 0x474E5554u : 0x47FF5554u, 0x43FF41FFu, x & 0x07070707u);
    return __builtin_amdgcn_sad_u8(x & 0xDFDFDFDFu, expect, acc);
}
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t suspect16(u32x4 q, uint32_t acc) {
    return suspect<ALLOW_N>(q.w, suspect<ALLOW_N>(q.z, suspect<ALLOW_N>(q.y, suspect<ALLOW_N>(q.x, acc))));
}
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t invalid16(u32x4 q) {  // exact: bytes of the vector outside the alphabet
    return __builtin_popcount(invalid_mask<ALLOW_N>(q.x)) + __builtin_popcount(invalid_mask<ALLOW_N>(q.y)) +
           __builtin_popcount(invalid_mask<ALLOW_N>(q.z)) + __builtin_popcount(invalid_mask<ALLOW_N>(q.w));
}
// the same with only bytes [lo, hi) of the vector counted (0 <= lo, hi <= 16; the window kernels' first and last rows):
// every other byte is replaced by 'A', a letter
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t invalid16_range(u32x4 q, int lo, int hi) {
    uint32_t c = 0;
    const uint32_t d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int a = lo - 4 * j, b = hi - 4 * j;  // bytes [a, b) of this dword are counted
        const uint32_t from = a <= 0 ? 0xFFFFFFFFu : a >= 4 ? 0u : 0xFFFFFFFFu << (8 * a);
        const uint32_t upto = b >= 4 ? 0xFFFFFFFFu : b <= 0 ? 0u : ~(0xFFFFFFFFu << (8 * b));
        const uint32_t keep = from & upto;
        c += __builtin_popcount(invalid_mask<ALLOW_N>((d[j] & keep) | (0x41414141u & ~keep)));
    }
    return c;
}
// one no-return atomic per wave that has something to add (none on clean data)
__device__ __forceinline__ void wave_sum_to(uint64_t v, unsigned long long* dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)v, off, 64), hi = __shfl_down((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0 && v) (void)__hip_atomic_fetch_add(dst, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Where a checked launch adds: ONE u64 (mask 0), or -- CNT_SPREAD_COUNT -- CNT_COUNT_SLOTS of them, workgroup b adding to slot
// b & mask.  Atomics on one address serialise at ~12 ns apiece wherever they come from: 2^34 nt in which every 2-KiB tile holds a
// stray (a FASTA file with its line feeds) is 8.4 M of them = 100 ms against a 3.1-ms encode -- and they serialise per cache LINE,
// not per address (256 slots = 16 lines: 5.8 ms), hence 2048 slots = 128 lines.
struct BadCounter {
    unsigned long long* p = nullptr;
    uint32_t mask = 0;
    BadCounter() = default;
    BadCounter(unsigned long long* q, uint32_t m = 0) : p(q), mask(m) {}
    explicit operator bool() const { return p != nullptr; }
    BadCounter slot(int i) const { return BadCounter(p ? p + (size_t)i * (mask + 1) : nullptr, mask); }
};
__device__ __forceinline__ void wave_add_invalid(uint32_t bad, unsigned long long* dst) {
    if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) wave_sum_to(bad, dst);
}

// ---------------------------------------------------------------------------
// decode arithmetic: one packed byte -> 4 ASCII bytes (one dword)
// ---------------------------------------------------------------------------
// b | b<<6 | b<<12 | b<<18 puts code k at bits 8k..8k+1 (the spread the
// reference does with a carry-less multiply, n_to_bits.rs:357-365,381-384);
// v_perm_b32 is then a 4-entry byte LUT: code -> "ACTG" (n_to_bits.rs:23-30).
// Cost matters here, unlike in encode: the 1:4 stream is bound by the lifetime of its resident waves, and the
// arithmetic-scaling probes (bench/probes.hip kinds 11..18, profiles/r03_ab_arith_*.jsonl) show decode's time growing
// with every VALU instruction between the load's return and the store (x1 = +2.2 % over the arithmetic-free stream =
// exactly this kernel's gap to its probe; encode is flat up to x4).  So the spread is spelled to need 4 instructions
// per output dword instead of 5: the byte extraction rides in the SDWA operand select of a v_mul_u32_u24 by 0x1001
// (b | b<<12: the two copies are 12 >= 8 bits apart, so the product has no carries), one v_lshl_or_b32 by 6 brings
// codes 1 and 3 to bits 8..9 and 24..25, one v_and_b32 masks, v_perm_b32 looks up.  (A single multiply b * 0x41041 is
// NOT the same thing: copies 6 bits apart overlap, and where OR ignores the overlap ADD carries into the bits the
// mask keeps -- tried, 104 GPU tests said no.)
__device__ __forceinline__ uint32_t dec1(uint32_t b /* 0..255 */) {
#ifdef CNT_DEC_SHIFT_OR  // rounds 1-2: extract, two shift-ORs, mask (5 VALU per output dword)
    uint32_t t = (b << 6) | b;
    uint32_t sel = ((t << 12) | t) & 0x03030303u;
#else
    uint32_t u = __umul24(b, 0x1001u);             // b at bits 0..7 and 12..19
    uint32_t sel = ((u << 6) | u) & 0x03030303u;   // codes at bits 0..1, 8..9, 16..17, 24..25
#endif
    return __builtin_amdgcn_perm(0u, 0x47544341u /* 'A','C','T','G' = bytes 0..3 */, sel);
}

__device__ __forceinline__ u32x4 dec4(uint32_t x) {  // 4 packed bytes -> 16 ASCII bytes
    u32x4 r;
    r.x = dec1(x & 0xFFu);
    r.y = dec1((x >> 8) & 0xFFu);
    r.z = dec1((x >> 16) & 0xFFu);
    r.w = dec1(x >> 24);
    return r;
}

// Residency cap.  The launchers may pass a dynamic-LDS size whose only purpose is to limit
// how many workgroups fit on a CU (160 KiB / size): with ~24 resident waves per CU instead
// of 32 the set of tiles in flight is a tighter address window and HBM runs 1-3 % faster
// (bench/tune_lab5.hip).  The kernels never use the memory; this never-taken store only
// keeps the allocation attached to them.
extern __shared__ uint32_t residency_pad[];
__device__ __forceinline__ void touch_residency_pad(uint32_t n_tiles, uint32_t v) {
    if (n_tiles == ~0u) residency_pad[threadIdx.x] = v;  // a launch has < 2^26 tiles
}

// ===========================================================================
// EDGES.  What the tile kernels cannot take -- the head words a launcher peels to line-align the
// stores, and the ragged end behind the last whole tile -- used to be one or two extra launches of
// the generic kernels (~5 us each behind a 0.2 ms kernel at 1 GiB: 2.5 % at BASELINE.json's
// configs[1] size).  They now ride in the SAME launch, in the same grid: the LAST `groups` workgroups
// of the launch, after they have issued their own tile's stores, also run the byte-granular body over
// the edge items (a wave-uniform scalar branch at the END of the kernel: nothing is added in front of
// a tile's loads, and the edge struct's kernel arguments are only fetched by the workgroups that use
// them -- a first version with extra workgroups and the branch at the top made every launch start
// with two dependent kernarg fetches and cost an isolated 0.2 ms launch ~1 us).
// ===========================================================================
constexpr uint64_t kNoLutWord = ~0ull;

// One output word from byte loads, any alignment, any length; zero-pads the last word
// (n_to_bits.rs:35).  `lut` = BYTE_LUT semantics for THIS word (CNT_STRICT_LUT everywhere, or
// CNT_TAIL_LUT on the final partial word -- where the reference's SIMD encoders call
// n_to_bits_lut, n_to_bits.rs:109-111,160-162,201-203,253-255).
__device__ __forceinline__ uint64_t encode_word_bytes(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t w, bool lut) {
    const uint64_t i0 = w << 5;
    uint64_t acc = 0;
    const int m = (n_len - i0) < 32 ? (int)(n_len - i0) : 32;
    for (int k = 0; k < m; k += 4) {
        uint32_t x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (k + j < m) x |= (uint32_t)n[i0 + k + j] << (8 * j);
        // a byte that was never loaded is 0 -> code 0 in both modes
        if (lut) x = strict_filter(x);
        acc |= (uint64_t)__builtin_amdgcn_ubfe(enc_gather(x & 0x06060606u), 19, 8) << (2 * k);
    }
    return acc;
}
// the same, also counting the word's bytes outside ACGTUacgtu into `bad` (bytes that do not exist count as 'A', a letter)
__device__ __forceinline__ uint64_t encode_word_bytes_checked(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t w, bool lut, uint32_t& bad) {
    const uint64_t i0 = w << 5;
    uint64_t acc = 0;
    const int m = (n_len - i0) < 32 ? (int)(n_len - i0) : 32;
    for (int k = 0; k < m; k += 4) {
        uint32_t x = 0, pad = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (k + j < m) x |= (uint32_t)n[i0 + k + j] << (8 * j);
            else pad |= 0x41u << (8 * j);
        }
        bad += __builtin_popcount(invalid_mask<false>(x | pad));
        if (lut) x = strict_filter(x);
        acc |= (uint64_t)__builtin_amdgcn_ubfe(enc_gather(x & 0x06060606u), 19, 8) << (2 * k);
    }
    return acc;
}

// words [0, head_words) and [tail_first, words) of an encode, all relative to the caller's pointers
struct EncodeEdges {
    const uint8_t* n;
    uint64_t* out;
    uint64_t n_len, head_words, tail_first, words;
    uint64_t lut_from;  // words >= lut_from take BYTE_LUT semantics (kNoLutWord: none)
    uint32_t groups;    // the last `groups` workgroups of the launch share the edge items (0: this launch has none)
};
template <bool STRICT>
__device__ __forceinline__ void encode_edges(const EncodeEdges& e, uint64_t idx, uint64_t stride) {
    const uint64_t items = e.head_words + (e.words - e.tail_first);
    for (uint64_t i = idx; i < items; i += stride) {
        const uint64_t w = i < e.head_words ? i : e.tail_first + (i - e.head_words);
        e.out[w] = encode_word_bytes(e.n, e.n_len, w, STRICT || w >= e.lut_from);
    }
}
template <bool STRICT>
__device__ __forceinline__ uint32_t encode_edges_checked(const EncodeEdges& e, uint64_t idx, uint64_t stride) {  // returns this thread's count
    const uint64_t items = e.head_words + (e.words - e.tail_first);
    uint32_t bad = 0;
    for (uint64_t i = idx; i < items; i += stride) {
        const uint64_t w = i < e.head_words ? i : e.tail_first + (i - e.head_words);
        e.out[w] = encode_word_bytes_checked(e.n, e.n_len, w, STRICT || w >= e.lut_from, bad);
    }
    return bad;
}
inline uint64_t encode_edge_items(const EncodeEdges& e) { return e.head_words + (e.words - e.tail_first); }

// nucleotides [0, head) and [tail_lo, len) of a decode.  The head exists because the output pointer is NOT aligned -- but
// only its first <= 15 letters are: from the first 16-B boundary of the output on, head and tail are spelled 16 letters per
// work item (two packed dwords through v_alignbit_b32, one 16-B store), and only the few letters in front of that boundary
// and behind the last whole 16 go one per item.  (Round 5: a head of up to 4 095 + 4 x 4 096 letters -- the peel to a page,
// the pages that place the XCD turns, the pages that keep the window kernel's reads inside the buffer -- was up to 20 479
// byte items shared by 4 096 threads: microseconds behind a 10-40 us kernel, -2 % per page at 2^26-2^28 nt.)
struct DecodeEdges {
    const uint64_t* bits;
    uint8_t* out;
    uint64_t head, tail_lo, len;
    uint32_t groups;  // as in EncodeEdges
};
// how the edge items are laid out: a byte items, nv 16-letter items of the head, nvt of the tail, rb byte items behind them
struct DecodeEdgeLayout {
    uint64_t a, nv, nvt, rb;
};
__host__ __device__ __forceinline__ DecodeEdgeLayout decode_edge_layout(const DecodeEdges& e) {
    DecodeEdgeLayout l;
    const uint64_t to16 = (16 - (reinterpret_cast<uintptr_t>(e.out) & 15)) & 15;
    l.a = e.head < to16 ? e.head : to16;
    l.nv = (e.head - l.a) >> 4;
    if ((e.head - l.a) & 15) {  // a head that does not end on a 16-B boundary of the output (never the launchers'): all bytes
        l.a = e.head;
        l.nv = 0;
    }
    const uint64_t tail = e.len - e.tail_lo;
    const bool tail_vec = ((reinterpret_cast<uintptr_t>(e.out) + e.tail_lo) & 15) == 0;
    l.nvt = tail_vec ? tail >> 4 : 0;
    l.rb = tail - (l.nvt << 4);
    return l;
}
__device__ __forceinline__ void decode_edge_letter(const DecodeEdges& e, uint64_t i) {
    const uint32_t code = (uint32_t)(e.bits[i >> 5] >> ((i & 31) << 1)) & 3u;
    e.out[i] = (uint8_t)(0x47544341u >> (code << 3));  // "ACTG"[code], n_to_bits.rs:23-30
}
// 16 letters from nucleotide n0 on (out + n0 on a 16-B boundary, n0 + 16 <= len): the second dword is only touched when it
// holds some of these letters' bits, so nothing behind the caller's `len` is read
__device__ __forceinline__ void decode_edge_vector(const DecodeEdges& e, uint64_t n0) {
    const uint32_t* dw = reinterpret_cast<const uint32_t*>(e.bits) + (n0 >> 4);
    const uint32_t sh = 2u * (uint32_t)(n0 & 15);
    const uint32_t lo = dw[0], hi = sh ? dw[1] : 0u;
    *reinterpret_cast<u32x4*>(e.out + n0) = dec4(__builtin_amdgcn_alignbit(hi, lo, sh));
}
__device__ __forceinline__ void decode_edges(const DecodeEdges& e, uint64_t idx, uint64_t stride) {
    const DecodeEdgeLayout l = decode_edge_layout(e);
    const uint64_t head_first_vec = l.a;
    const uint64_t items = l.a + l.nv + l.nvt + l.rb;
    for (uint64_t k = idx; k < items; k += stride) {
        if (k < l.a) decode_edge_letter(e, k);
        else if (k < l.a + l.nv) decode_edge_vector(e, head_first_vec + ((k - l.a) << 4));
        else if (k < l.a + l.nv + l.nvt) decode_edge_vector(e, e.tail_lo + ((k - l.a - l.nv) << 4));
        else decode_edge_letter(e, e.tail_lo + (l.nvt << 4) + (k - l.a - l.nv - l.nvt));
    }
}
inline uint64_t decode_edge_items(const DecodeEdges& e) {
    const DecodeEdgeLayout l = decode_edge_layout(e);
    return l.a + l.nv + l.nvt + l.rb;
}

// the ragged end of a fused round trip: words [tail_first, words) with their letters -- a thread packs one word from
// byte loads and spells the same codes back out (no thread waits for another's word)
struct RoundTripEdges {
    const uint8_t* n;
    uint64_t* packed;
    uint8_t* back;
    uint64_t n_len, tail_first, words, lut_from;
    uint32_t groups;
};
template <bool STRICT>
__device__ __forceinline__ void round_trip_edges(const RoundTripEdges& e, uint64_t idx, uint64_t stride) {
    for (uint64_t w = e.tail_first + idx; w < e.words; w += stride) {
        const uint64_t acc = encode_word_bytes(e.n, e.n_len, w, STRICT || w >= e.lut_from);
        e.packed[w] = acc;
        const uint64_t i0 = w << 5;
        const int m = (e.n_len - i0) < 32 ? (int)(e.n_len - i0) : 32;
        for (int k = 0; k < m; ++k) e.back[i0 + k] = (uint8_t)(0x47544341u >> ((uint32_t)((acc >> (2 * k)) & 3u) << 3));
    }
}

template <bool STRICT>
__device__ __forceinline__ uint32_t round_trip_edges_checked(const RoundTripEdges& e, uint64_t idx, uint64_t stride) {
    uint32_t bad = 0;
    for (uint64_t w = e.tail_first + idx; w < e.words; w += stride) {
        const uint64_t acc = encode_word_bytes_checked(e.n, e.n_len, w, STRICT || w >= e.lut_from, bad);
        e.packed[w] = acc;
        const uint64_t i0 = w << 5;
        const int m = (e.n_len - i0) < 32 ? (int)(e.n_len - i0) : 32;
        for (int k = 0; k < m; ++k) e.back[i0 + k] = (uint8_t)(0x47544341u >> ((uint32_t)((acc >> (2 * k)) & 3u) << 3));
    }
    return bad;
}

// the end of every tile kernel: the launch's last e.groups workgroups (groups <= n_tiles) share the edge items
#define CNT_ENCODE_EDGES_TAIL(BLOCK_)                                                                                   \
    if (blockIdx.x + e.groups >= n_tiles)                                                                               \
        encode_edges<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * (BLOCK_) + threadIdx.x, (uint64_t)e.groups * (BLOCK_));
#define CNT_DECODE_EDGES_TAIL(BLOCK_)                                                                                   \
    if (blockIdx.x + e.groups >= n_tiles)                                                                               \
        decode_edges(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * (BLOCK_) + threadIdx.x, (uint64_t)e.groups * (BLOCK_));
// (Round 5 also tried the launch's FIRST workgroups for the edge items, so that their one extra trip to memory overlaps with the
// rest of the grid instead of ending it: +-1-4 % either way at 2^22-2^26 nt, +-0.2 % from 2^28 on -- noise; not adopted.)

// ===========================================================================
// ENCODE.  One workgroup = one tile of BLOCK*U*16 nt, no loop: the launch has one
// workgroup per whole tile plus the edge workgroups above.
// ===========================================================================

// STREAM: 16-B loads (1 KiB per wave-instruction, coalesced) -> one packed dword
// per lane per load, stored 4 B per lane (256 B per wave-instruction).
// (`bad_out` / CHECK: the *_checked twin below counts the tile's bytes outside the alphabet behind its stores)
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT, bool CHECK>
__device__ __forceinline__ void n_to_bits_stream_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs,
                                                      const EncodeEdges& e, unsigned long long* __restrict__ bad_out) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
    for (int u = 0; u < U; ++u)
        __builtin_amdgcn_raw_buffer_store_b32(enc16<STRICT>(v[u]), rout, (u * BLOCK + tid) * 4, 0, SAUX);
    if constexpr (CHECK) {
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) sus = suspect16<false>(v[u], sus);
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {  // never on clean data
#pragma unroll
            for (int u = 0; u < U; ++u) bad += invalid16<false>(v[u]);
        }
        if (blockIdx.x + e.groups >= n_tiles)
            bad += encode_edges_checked<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * BLOCK + tid, (uint64_t)e.groups * BLOCK);
        wave_add_invalid(bad, bad_out);
    } else {
        CNT_ENCODE_EDGES_TAIL(BLOCK)
    }
}
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(BLOCK) void n_to_bits_stream(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          uint32_t n_tiles, uint32_t xs, EncodeEdges e) {
    n_to_bits_stream_body<BLOCK, U, C, LAUX, SAUX, STRICT, false>(in, out, n_tiles, xs, e, nullptr);
}
// CHECKED (round 6; cnt_n_to_bits_checked_dev): the same tile, and *bad += the number of its bytes outside ACGTUacgtu.
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(BLOCK) void n_to_bits_stream_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                                  uint32_t n_tiles, uint32_t xs, EncodeEdges e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    n_to_bits_stream_body<BLOCK, U, C, LAUX, SAUX, STRICT, true>(in, out, n_tiles, xs, e, bad + (blockIdx.x & slot_mask));
}

// WINDOW: variant 0's one-wave shape with U loads per lane (U = 4: 4 KiB in, 1 KiB out -- the aligned kernel loses 0.6 % to
// U = 2 at that size, profiles/r03_ab_step_encode_policies_plain_order.log, and the window's read-ahead line is 3 % of a
// 4-KiB tile's reads instead of 6 % of a 2-KiB tile's) for an input that starts at ANY
// byte address.  The wave loads the 128-B-aligned window that covers its tile -- `in` is the
// caller's pointer rounded down to a line, `phase` (1..127) the bytes dropped -- so every load
// is a whole-line access exactly as in the aligned kernel; each lane packs the ALIGNED 16 bytes
// it loaded, and the byte phase is then applied to the packed codes, which are 4x smaller:
// output dword c = bits [2*(phase%16) ..) of code dword c + phase/16 and its successor, fetched
// from the wave's LDS slab and funnel-shifted (v_alignbit_b32).  The slab is the dynamic-LDS
// allocation that already caps residency (>= 768 B; one wave per workgroup, so it is private).
// A tile reads up to 127 B before and 144 B behind itself; the launcher keeps both inside the
// caller's buffer.  (The OUTPUT side cannot be treated this way -- a store stream that is not
// 64-B aligned costs ~30 %, profiles/r01_align_lab*.json -- so the launcher peels head words until
// the stores are line-aligned and hands the resulting input phase here.)
template <int U, int C, int LAUX, int SAUX, bool STRICT, bool CHECK>
__device__ __forceinline__ void n_to_bits_window_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t phase,
                                                      uint32_t xs, const EncodeEdges& e, unsigned long long* __restrict__ bad_out) {
    constexpr uint32_t TILE_IN = kWave * U * 16, TILE_OUT = TILE_IN / 4, SLACK = 144;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN + SLACK);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t lane = threadIdx.x;
    const uint32_t q = phase >> 4, sh = (phase & 15) << 1;
    // lanes 0..q fetch the q+1 vectors behind the tile; the others aim past the descriptor's range,
    // which returns 0 without touching memory
    const uint32_t off2 = lane <= q ? (U * kWave + lane) * 16 : 0xFFFFFF00u;
    u32x4 v[U + 1];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * kWave + lane) * 16, 0, LAUX));
    v[U] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, off2, 0, LAUX));
#pragma unroll
    for (int u = 0; u <= U; ++u) residency_pad[u * kWave + lane] = enc16<STRICT>(v[u]);
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t o = __builtin_amdgcn_alignbit(residency_pad[u * kWave + lane + q + 1], residency_pad[u * kWave + lane + q], sh);
        __builtin_amdgcn_raw_buffer_store_b32(o, rout, (u * kWave + lane) * 4, 0, SAUX);
    }
    if constexpr (CHECK) {
        // the tile's OWN letters are window bytes [phase, TILE_IN + phase).  The fast path looks at every byte the wave loaded
        // (a neighbour's bad byte sends this wave to the exact count for nothing: harmless); the exact count takes the first
        // row from byte `phase` on and the read-ahead row up to it, so that every byte of the call is counted by ONE tile.
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) sus = suspect16<false>(v[u], sus);
        sus += lane <= q ? suspect16<false>(v[U], 0u) : 0u;  // lanes behind q hold the descriptor's zeros, not bytes
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {
            bad = invalid16_range<false>(v[0], (int)phase - 16 * (int)lane, 16) + invalid16_range<false>(v[U], 0, (int)phase - 16 * (int)lane);
#pragma unroll
            for (int u = 1; u < U; ++u) bad += invalid16<false>(v[u]);
        }
        if (blockIdx.x + e.groups >= n_tiles)
            bad += encode_edges_checked<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * kWave + lane, (uint64_t)e.groups * kWave);
        wave_add_invalid(bad, bad_out);
    } else {
        CNT_ENCODE_EDGES_TAIL(kWave)
    }
}
template <int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(kWave) void n_to_bits_window(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          uint32_t n_tiles, uint32_t phase, uint32_t xs, EncodeEdges e) {
    n_to_bits_window_body<U, C, LAUX, SAUX, STRICT, false>(in, out, n_tiles, phase, xs, e, nullptr);
}
template <int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(kWave) void n_to_bits_window_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles,
                                                                  uint32_t phase, uint32_t xs, EncodeEdges e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    n_to_bits_window_body<U, C, LAUX, SAUX, STRICT, true>(in, out, n_tiles, phase, xs, e, bad + (blockIdx.x & slot_mask));
}

// FUSED round trip (BASELINE.json configs[3]): one pass that reads the ASCII once and writes BOTH the
// packed words and the decoded (canonical: upper case, U -> T) ASCII -- the packed dword a lane just
// built is decoded from registers, so the packed form is never read back: 1 + 0.25 + 1 = 2.25 B/nt
// instead of the 2.5 B/nt of encode followed by decode.  Launched as <64, 4, 1>: one wave, four loads
// per lane, 4 KiB of ASCII per workgroup, plain dispatch order (codec2_launch.hpp, bench/tune_lab11.hip).
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT, bool CHECK>
__device__ __forceinline__ void round_trip_stream_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back,
                                                       uint32_t n_tiles, uint32_t xs, const RoundTripEdges& e, unsigned long long* __restrict__ bad_out) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_PK = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rpk = rsrc_of(packed + t * TILE_PK, TILE_PK);
    const __amdgpu_buffer_rsrc_t rbk = rsrc_of(back + t * TILE_IN, TILE_IN);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t code = enc16<STRICT>(v[u]);
        __builtin_amdgcn_raw_buffer_store_b32(code, rpk, (u * BLOCK + tid) * 4, 0, SAUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(code)), rbk, (u * BLOCK + tid) * 16, 0, SAUX);
    }
    if constexpr (CHECK) {
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) sus = suspect16<false>(v[u], sus);
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) bad += invalid16<false>(v[u]);
        }
        if (blockIdx.x + e.groups >= n_tiles)
            bad += round_trip_edges_checked<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * BLOCK + tid, (uint64_t)e.groups * BLOCK);
        wave_add_invalid(bad, bad_out);
    } else {
        if (blockIdx.x + e.groups >= n_tiles)
            round_trip_edges<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * BLOCK + tid, (uint64_t)e.groups * BLOCK);
    }
}
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(BLOCK) void round_trip_stream(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed,
                                                           uint8_t* __restrict__ back, uint32_t n_tiles, uint32_t xs, RoundTripEdges e) {
    round_trip_stream_body<BLOCK, U, C, LAUX, SAUX, STRICT, false>(in, packed, back, n_tiles, xs, e, nullptr);
}
template <int BLOCK, int U, int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(BLOCK) void round_trip_stream_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back,
                                                                   uint32_t n_tiles, uint32_t xs, RoundTripEdges e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    round_trip_stream_body<BLOCK, U, C, LAUX, SAUX, STRICT, true>(in, packed, back, n_tiles, xs, e, bad + (blockIdx.x & slot_mask));
}

// FUSED round trip at ANY alignment of its three pointers, still one pass and one launch (round 4).  The tile is laid
// on the WIDE store stream: tile t covers nucleotides [t0 + 4096 t, +4096) with d_back + t0 on a 128-B line (a 4-KiB
// boundary for large buffers), so the 4 KiB of decoded ASCII leave as the aligned kernel's stores do.  The other two
// streams take whatever phase that leaves them, and both are resolved on the PACKED CODES, which are 4x smaller than the
// text (n_to_bits_window's trick, twice):
//   * loads: the wave reads the 128-B-aligned window that covers its tile (`in` = d_n + t0 rounded down to a line,
//     `phase` = the 0..127 bytes dropped, nine vectors of slack behind the tile), every lane packs the ALIGNED 16 bytes
//     it loaded, and the code dwords go to the wave's LDS slab;
//   * decoded stream: the 16 letters a lane stores come from code bits [2 * phase ..) of the slab -- two adjacent slab
//     dwords funnel-shifted (v_alignbit_b32) -- and are spelled out from registers as in the aligned kernel;
//   * packed stream: memory dword m holds nucleotides [16 m, 16 m + 16) of the CALLER's numbering; t0 is not a multiple
//     of 16 in general, and the dword behind it is not on a 64-B segment of d_bits in general -- and packed stores that
//     split 64-B segments between store instructions cost 11-18 % of the whole kernel (partial-segment writes,
//     profiles/r04_align_round_trip_first.jsonl).  So tile t writes the 256 dwords that start at p0 + 256 t, p0 = the
//     segment-aligned dword nearest to t0 (in front of it or behind it, at most 128 nucleotides away): a second funnel
//     read of the same slab at its own phase.  The window starts at the line that holds the earlier of the two first
//     nucleotides and carries up to 17 vectors of slack behind the tile for the later one.
// The slab is the dynamic-LDS allocation that caps residency (one wave per workgroup: it is private).  Everything in front
// of t0, behind the last tile, and the packed dword that straddles either border rides in the same launch as dword-granular
// edge items (round_trip_edges_any below); the launcher keeps the window's reads inside the caller's buffer.
struct RoundTripEdgesAny {
    const uint8_t* n;
    uint32_t* packed;  // the u64 words as dwords: dword d holds nucleotides [16 d, 16 d + 16)
    uint8_t* back;
    uint64_t n_len;
    uint64_t t0, t1;    // the tiles write letters [t0, t1) ...
    uint64_t p0, p1;    // ... and packed dwords [p0, p1), |16 p0 - t0| <= 128 + 15, p1 - p0 == (t1 - t0) / 16
    uint64_t dwords;    // 2 x words: the zero-padded upper half of a last word with <= 16 nucleotides is written too
    uint64_t lut_from;  // words >= lut_from take BYTE_LUT semantics (kNoLutWord: none)
    uint32_t groups;
};
// CHECK: returns the number of bytes outside ACGTUacgtu among the letters the TILES do not own (i < t0 or i >= t1: exactly
// the letters this body spells out), so that tiles + edge items count every byte of the call once
template <bool STRICT, bool CHECK = false>
__device__ __forceinline__ uint32_t round_trip_edges_any(const RoundTripEdgesAny& e, uint64_t idx, uint64_t stride) {
    // dwords [0, h) and [f, dwords) hold a letter or a packed dword that the tiles do not write
    const uint64_t h = max(e.p0, (e.t0 + 15) >> 4), f = min(e.p1, e.t1 >> 4);
    const uint64_t items = h + (e.dwords - f);
    uint32_t bad = 0;
    for (uint64_t k = idx; k < items; k += stride) {
        const uint64_t d = k < h ? k : f + (k - h);
        const uint64_t i0 = d << 4;
        const int m = i0 >= e.n_len ? 0 : ((e.n_len - i0) < 16 ? (int)(e.n_len - i0) : 16);
        const bool lut = STRICT || (d >> 1) >= e.lut_from;
        uint32_t code = 0;
        for (int q = 0; q < m; q += 4) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q + j < m) x |= (uint32_t)e.n[i0 + q + j] << (8 * j);
            if constexpr (CHECK) {
                uint32_t keep = 0;  // 0xFF per byte that exists and lies outside the tiles' letters
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (q + j < m && (i0 + q + j < e.t0 || i0 + q + j >= e.t1)) keep |= 0xFFu << (8 * j);
                bad += __builtin_popcount(invalid_mask<false>((x & keep) | (0x41414141u & ~keep)));
            }
            if (lut) x = strict_filter(x);
            code |= __builtin_amdgcn_ubfe(enc_gather(x & 0x06060606u), 19, 8) << (2 * q);
        }
        if (d < e.p0 || d >= e.p1) e.packed[d] = code;  // the tiles own [p0, p1)
        for (int q = 0; q < m; ++q) {
            const uint64_t i = i0 + q;
            if (i < e.t0 || i >= e.t1) e.back[i] = (uint8_t)(0x47544341u >> (((code >> (2 * q)) & 3u) << 3));  // "ACTG"[code]
        }
    }
    return bad;
}
constexpr uint32_t kRoundTripAnyTile = 64 * 4 * 16;
constexpr uint32_t kRoundTripAnySlackVecs = 25;  // 16-B vectors of the window a tile may read behind its 4 KiB (the launcher needs <= 24: phases < 128 + 256)
constexpr uint32_t kRoundTripAnySlack = kRoundTripAnySlackVecs * 16;  // bytes a tile's window may read behind the tile's own end
constexpr uint32_t kRoundTripAnySlab = 5 * 64 * 4;  // LDS bytes: four rows of code dwords + the slack row
template <int C, int LAUX, int SAUX, bool STRICT, bool CHECK>
__device__ __forceinline__ void round_trip_window_body(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back, uint32_t n_tiles,
                                                       uint32_t phase, uint32_t phase2, uint32_t xs, const RoundTripEdgesAny& e, unsigned long long* __restrict__ bad_out) {
    constexpr uint32_t TILE = kRoundTripAnyTile;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE, TILE + kRoundTripAnySlack);
    const __amdgpu_buffer_rsrc_t rpk = rsrc_of(packed + t * (TILE / 4), TILE / 4);
    const __amdgpu_buffer_rsrc_t rbk = rsrc_of(back + t * TILE, TILE);
    const uint32_t lane = threadIdx.x;
    const uint32_t q = phase >> 4, sh = (phase & 15) << 1;
    const uint32_t q2 = phase2 >> 4, sh2 = (phase2 & 15) << 1;  // both phases <= 127 + 143
    u32x4 v[5];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * kWave + lane) * 16, 0, LAUX));
    // the two funnels reach code dword 255 + q + 1 (+ q2), and only with a bit phase: ceil(max(phase, phase2) / 16) vectors
    // behind the tile's own 256, fetched by that many lanes; the others aim past the descriptor's range (zeros, no memory
    // access).  A phase of up to 128 letters is ONE further line.
    v[4] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane < ((max(phase, phase2) + 15u) >> 4) ? (4 * kWave + lane) * 16 : 0xFFFFFF00u, 0, LAUX));
#pragma unroll
    for (int u = 0; u < 5; ++u) residency_pad[u * kWave + lane] = enc16<STRICT>(v[u]);
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t j = u * kWave + lane;
        const uint32_t cb = __builtin_amdgcn_alignbit(residency_pad[j + q + 1], residency_pad[j + q], sh);      // letters [t0 + 16 j, +16)
        const uint32_t cp = __builtin_amdgcn_alignbit(residency_pad[j + q2 + 1], residency_pad[j + q2], sh2);  // packed dword p0 + j
        __builtin_amdgcn_raw_buffer_store_b32(cp, rpk, j * 4, 0, SAUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(cb)), rbk, j * 16, 0, SAUX);
    }
    if constexpr (CHECK) {
        // the tile OWNS the letters it spells out: window bytes [phase, TILE + phase) (n_to_bits_window_body's rule)
        uint32_t sus = 0, bad = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) sus = suspect16<false>(v[u], sus);
        sus += lane < ((max(phase, phase2) + 15u) >> 4) ? suspect16<false>(v[4], 0u) : 0u;
        if (__builtin_amdgcn_ballot_w64(sus != 0) != 0) {
            bad = invalid16_range<false>(v[0], (int)phase - 16 * (int)lane, 16) + invalid16_range<false>(v[4], 0, (int)phase - 16 * (int)lane);
#pragma unroll
            for (int u = 1; u < 4; ++u) bad += invalid16<false>(v[u]);
        }
        if (blockIdx.x + e.groups >= n_tiles)
            bad += round_trip_edges_any<STRICT, true>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * kWave + lane, (uint64_t)e.groups * kWave);
        wave_add_invalid(bad, bad_out);
    } else {
        if (blockIdx.x + e.groups >= n_tiles)
            round_trip_edges_any<STRICT>(e, (uint64_t)(blockIdx.x + e.groups - n_tiles) * kWave + lane, (uint64_t)e.groups * kWave);
    }
}
template <int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(kWave) void round_trip_window(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back,
                                                          uint32_t n_tiles, uint32_t phase, uint32_t phase2, uint32_t xs, RoundTripEdgesAny e) {
    round_trip_window_body<C, LAUX, SAUX, STRICT, false>(in, packed, back, n_tiles, phase, phase2, xs, e, nullptr);
}
template <int C, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(kWave) void round_trip_window_checked(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back, uint32_t n_tiles,
                                                                  uint32_t phase, uint32_t phase2, uint32_t xs, RoundTripEdgesAny e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    round_trip_window_body<C, LAUX, SAUX, STRICT, true>(in, packed, back, n_tiles, phase, phase2, xs, e, bad + (blockIdx.x & slot_mask));
}
// inputs shorter than a tile + slack: the edge body alone, one launch
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void round_trip_generic(RoundTripEdgesAny e) {
    round_trip_edges_any<STRICT>(e, blockIdx.x * (uint64_t)kBlock + threadIdx.x, (uint64_t)gridDim.x * kBlock);
}
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void round_trip_generic_checked(RoundTripEdgesAny e, unsigned long long* __restrict__ bad, uint32_t slot_mask) {
    wave_add_invalid(round_trip_edges_any<STRICT, true>(e, blockIdx.x * (uint64_t)kBlock + threadIdx.x, (uint64_t)gridDim.x * kBlock), bad + (blockIdx.x & slot_mask));
}

// LDS (kept as the measured alternative): each wave loads U x 1 KiB coalesced,
// packs to U dwords per lane, parks them in its private LDS slab in output order
// (ds_write_b32, conflict-free), reads back 16 B per lane (ds_read_b128) and
// stores U/4 coalesced 16-B vectors.  No block barrier: the slab is per wave.
template <int BLOCK, int U, int LAUX, int SAUX, bool STRICT>
__global__ __launch_bounds__(BLOCK) void n_to_bits_lds(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       uint32_t n_tiles, EncodeEdges e) {
    static_assert(U % 4 == 0, "U must be a multiple of 4");
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    __shared__ __attribute__((aligned(16))) uint32_t slab[BLOCK * U];
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* my = slab + wave * (U * kWave);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, ((wave * U + u) * kWave + lane) * 16, 0, LAUX));
#pragma unroll
    for (int u = 0; u < U; ++u) my[u * kWave + lane] = enc16<STRICT>(v[u]);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < U / 4; ++j) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(my + (j * kWave + lane) * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, q), rout, (wave * U * kWave + (j * kWave + lane) * 4) * 4, 0, SAUX);
    }
    CNT_ENCODE_EDGES_TAIL(BLOCK)
}

// Generic: one thread per output word, byte loads, any alignment, any length; zero-pads the last
// word (n_to_bits.rs:35).  Small inputs and inputs shorter than one tile (everything else rides
// in the tile kernels' edge workgroups).
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_generic(const uint8_t* __restrict__ n, uint64_t n_len,
                                                            uint64_t* __restrict__ out, uint64_t first_word,
                                                            uint64_t n_words, uint64_t lut_from) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock)
        out[w] = encode_word_bytes(n, n_len, w, STRICT || w >= lut_from);
}

template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_generic_checked(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t* __restrict__ out, uint64_t first_word,
                                                                    uint64_t n_words, uint64_t lut_from, unsigned long long* __restrict__ bad_out, uint32_t slot_mask) {
    bad_out += blockIdx.x & slot_mask;
    uint32_t bad = 0;
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock)
        out[w] = encode_word_bytes_checked(n, n_len, w, STRICT || w >= lut_from, bad);
    wave_add_invalid(bad, bad_out);
}

// STAGED: the host tier's small-call path (hip/host_tier.inc).  The kernel reads the shim's own PINNED
// staging buffer over PCIe and writes the pinned result buffer: the staging base is 16-B aligned and the
// host zero-pads the input to a whole word, so a thread takes its 32 nt with two 16-B loads issued
// together -- ONE PCIe round trip per thread, where the generic kernel's byte loads chain several
// (bench/latency_lab.hip: 18.8 -> 15.7 us for a 40 000-nt call).  One thread per output word.
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_staged(const uint8_t* __restrict__ n, uint64_t* __restrict__ out, uint64_t n_words,
                                                           uint64_t lut_from) {
    const uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x;
    if (w >= n_words) return;
    const u32x4* p = reinterpret_cast<const u32x4*>(n + 32 * w);
    const u32x4 a = p[0], b = p[1];
    if (!STRICT && w >= lut_from)  // CNT_TAIL_LUT: the final partial word through BYTE_LUT (zero padding -> code 0 either way)
        out[w] = (uint64_t)enc16<true>(a) | ((uint64_t)enc16<true>(b) << 32);
    else
        out[w] = (uint64_t)enc16<STRICT>(a) | ((uint64_t)enc16<STRICT>(b) << 32);
}

// ... and its checked twin (cnt_n_to_bits_checked): the host pads the final word with 'A' -- code 0 like the zero byte, but a
// letter -- so every byte the kernel sees counts
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_staged_checked(const uint8_t* __restrict__ n, uint64_t* __restrict__ out, uint64_t n_words, uint64_t lut_from,
                                                                   unsigned long long* __restrict__ bad_out, uint32_t slot_mask) {
    bad_out += blockIdx.x & slot_mask;
    const uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x;
    uint32_t bad = 0;
    if (w < n_words) {
        const u32x4* p = reinterpret_cast<const u32x4*>(n + 32 * w);
        const u32x4 a = p[0], b = p[1];
        if (!STRICT && w >= lut_from) out[w] = (uint64_t)enc16<true>(a) | ((uint64_t)enc16<true>(b) << 32);
        else out[w] = (uint64_t)enc16<STRICT>(a) | ((uint64_t)enc16<STRICT>(b) << 32);
        bad = invalid16<false>(a) + invalid16<false>(b);
    }
    wave_add_invalid(bad, bad_out);
}

// ===========================================================================
// DECODE.  One workgroup = one tile of BLOCK*U*16 nt of OUTPUT.
// ===========================================================================

// STREAM: 4-B loads (256 B per wave-instruction) -> 16-B stores (1 KiB per
// wave-instruction), both coalesced.
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void bits_to_n_stream(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                          uint32_t n_tiles, uint32_t xs, DecodeEdges e) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    uint32_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4, 0, LAUX);
    touch_residency_pad(n_tiles, x[0]);
#pragma unroll
    for (int u = 0; u < U; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x[u])), rout, (u * BLOCK + tid) * 16, 0, SAUX);
    CNT_DECODE_EDGES_TAIL(BLOCK)
}

// SHIFTED: output tile-aligned, the packed stream entered at any even bit position.  `in` is
// the dword that holds the tile sequence's first nucleotide, `sh` = 2 * (its index in that
// dword's 16): a lane funnel-shifts two adjacent dwords (v_alignbit_b32) into its 32 bits.  The
// second dword holds bits of the lane's own last nucleotides (sh > 0), so nothing past the
// caller's `len` is read.  Used after the launcher has peeled head nucleotides to line-align
// the stores.
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void bits_to_n_shifted(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                           uint32_t n_tiles, uint32_t sh, uint32_t xs, DecodeEdges e) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN + 4);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    uint32_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint32_t lo = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4, 0, LAUX);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4 + 4, 0, LAUX);
        x[u] = __builtin_amdgcn_alignbit(hi, lo, sh);
    }
    touch_residency_pad(n_tiles, x[0]);
#pragma unroll
    for (int u = 0; u < U; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x[u])), rout, (u * BLOCK + tid) * 16, 0, SAUX);
    CNT_DECODE_EDGES_TAIL(BLOCK)
}

// WINDOW (round 5): output tile-aligned, the packed stream entered at ANY bit position -- off its 128-B lines included --
// with every global load line-aligned.  bits_to_n_stream's four 4-B loads per lane cover 256 B per wave-instruction; when
// the packed pointer is not on a line each of them straddles THREE lines instead of two (3-6 % of the kernel, whatever the
// XCD turns do: profiles/r05_decode_off_grid.md), and bits_to_n_shifted doubles them for the bit phase.  Here the wave
// loads the 128-B-aligned window over its tile's 1 KiB of packed words as ONE 16-B load per lane (eight whole lines per
// wave-instruction) plus, for the lanes that reach that far, the <= 8 vectors behind it (the others aim past the
// descriptor's range: zeros, no memory access -- branch-free, round_trip_window's trick), parks the dwords in its slab --
// the dynamic LDS that caps residency -- and every lane funnel-reads its four packed dwords at the stream's phase: dword
// phase q (0..31), bit phase sh (0..30, even).  `in` = the window of tile 0 (128-B aligned, inside the caller's buffer: the
// launcher sees to both ends).
constexpr uint32_t kWindowDecodeTile = 64 * 4 * 16;
constexpr uint32_t kWindowDecodeSlack = 9 * 16;        // bytes a tile may read behind its 1 KiB: (31 + 1 + 3) / 4 vectors, rounded up
constexpr uint32_t kWindowDecodeSlab = 2 * 64 * 4 * 4;  // LDS bytes: the tile's 256 dwords + one row behind them
template <int C, int LAUX, int SAUX>
__global__ __launch_bounds__(kWave) void bits_to_n_window(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t q,
                                                          uint32_t sh, uint32_t xs, DecodeEdges e) {
    constexpr uint32_t TILE_OUT = kWindowDecodeTile, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN + kWindowDecodeSlack);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t lane = threadIdx.x;
    // the funnels reach dword 255 + q, and 256 + q with a bit phase: ceil((q + (sh != 0)) / 4) vectors behind the tile's 64
    const uint32_t extra = (q + (sh ? 1u : 0u) + 3u) >> 2;
    const u32x4 v0 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, LAUX));
    const u32x4 v1 = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane < extra ? TILE_IN + lane * 16 : 0xFFFFFF00u, 0, LAUX));
    residency_pad[4 * lane + 0] = v0.x;
    residency_pad[4 * lane + 1] = v0.y;
    residency_pad[4 * lane + 2] = v0.z;
    residency_pad[4 * lane + 3] = v0.w;
    residency_pad[256 + 4 * lane + 0] = v1.x;
    residency_pad[256 + 4 * lane + 1] = v1.y;
    residency_pad[256 + 4 * lane + 2] = v1.z;
    residency_pad[256 + 4 * lane + 3] = v1.w;
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t j = u * kWave + lane + q;
        const uint32_t x = __builtin_amdgcn_alignbit(residency_pad[j + 1], residency_pad[j], sh);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x)), rout, (u * kWave + lane) * 16, 0, SAUX);
    }
    CNT_DECODE_EDGES_TAIL(kWave)
}

// LDS (measured alternative): each wave loads U/4 x 1 KiB of packed words as
// 16-B vectors, parks them in its LDS slab (ds_write_b128), re-reads dword
// (u*64+lane) (ds_read_b32, conflict-free) and stores U coalesced 16-B vectors.
template <int BLOCK, int U, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void bits_to_n_lds(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                       uint32_t n_tiles, DecodeEdges e) {
    static_assert(U % 4 == 0, "U must be a multiple of 4");
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    __shared__ __attribute__((aligned(16))) uint32_t slab[BLOCK * U];
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* my = slab + wave * (U * kWave);
    u32x4 q[U / 4];
#pragma unroll
    for (int j = 0; j < U / 4; ++j)
        q[j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (wave * U * kWave + (j * kWave + lane) * 4) * 4, 0, LAUX));
#pragma unroll
    for (int j = 0; j < U / 4; ++j) *reinterpret_cast<u32x4*>(my + (j * kWave + lane) * 4) = q[j];
    wave_lds_fence();
#pragma unroll
    for (int u = 0; u < U; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(my[u * kWave + lane])), rout,
                                               ((wave * U + u) * kWave + lane) * 16, 0, SAUX);
    CNT_DECODE_EDGES_TAIL(BLOCK)
}

// STAGED twin for the host tier's small-call path: one thread per packed word, 8-B load from the pinned staging
// buffer, all 32 letters written with two 16-B stores into the pinned result buffer (which has room for whole
// words; the host copies `len` bytes out of it).
__global__ __launch_bounds__(kBlock) void bits_to_n_staged(const uint64_t* __restrict__ bits, uint8_t* __restrict__ out, uint64_t n_words) {
    const uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x;
    if (w >= n_words) return;
    const uint64_t word = bits[w];
    u32x4* q = reinterpret_cast<u32x4*>(out + 32 * w);
    q[0] = dec4((uint32_t)word);
    q[1] = dec4((uint32_t)(word >> 32));
}

// completion flag of a small host-tier call: the last kernel of the call on its stream; the host spins on the
// pinned word instead of going through hipStreamSynchronize (3-4 us of an 18-us call)
__global__ void raise_flag(uint32_t* flag, uint32_t value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Generic: one thread per packed word, writes min(32, len - 32w) bytes with byte stores; any
// alignment.  Bits beyond `len` are ignored (n_to_bits.rs:60-65 stops at len).  Small inputs and
// inputs shorter than one tile (everything else rides in the tile kernels' edge workgroups).
__global__ __launch_bounds__(kBlock) void bits_to_n_generic(const uint64_t* __restrict__ bits, uint64_t len,
                                                            uint8_t* __restrict__ out, uint64_t first_word,
                                                            uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i0 = w << 5;
        const uint64_t word = bits[w];
        const int m = (len - i0) < 32 ? (int)(len - i0) : 32;
        for (int k = 0; k < m; k += 4) {
            uint32_t d = dec1((uint32_t)(word >> (2 * k)) & 0xFFu);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < m) out[i0 + k + j] = (uint8_t)(d >> (8 * j));
        }
    }
}

}  // namespace cnt
