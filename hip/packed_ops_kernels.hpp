// packed_ops_kernels.hpp -- operations on the 2-bit packed representation without decoding
// (SURVEY 8 f-4).  The reference does not implement these: its README only points at them
// ("Many operations (like Hamming distance) can be done directly on the bit strings without
// decoding", README.md:45; links to complement / hamming / validity code in another project,
// README.md:20-25) -- so there are no reference vectors and parity for this file is pinned
// only by the definitions restated in oracle/cnt_oracle.c (see oracle/README.md).
//
// Layout as everywhere: nucleotide i = bits 2*(i&31).. of word i>>5, codes A0 C1 T2 G3.
//   hamming            #{ i < len : code_a(i) != code_b(i) }
//   complement         A<->T, C<->G = flip the high bit of every 2-bit code (x ^ 0xAAAA...)
//   reverse complement out(i) = complement(in(len-1-i))
//   validate           #{ bytes of an ASCII buffer outside ACGTUacgtu (optionally also N/n) }
// All four are HBM streams.  Reductions write one partial sum per workgroup to a scratch array
// and a one-workgroup second kernel adds them up: one atomic per workgroup on a single counter
// made the Hamming kernel atomic-bound (5.4 -> 7.2 TB/s without them, bench/tune_lab7.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec2_kernels.hpp"

namespace cnt {

constexpr int kRedBlock = 1024;  // reduction kernels: big workgroups, few atomics (read-only streams like them)

__device__ __forceinline__ uint32_t diff_codes32(uint32_t a, uint32_t b) {  // # differing 2-bit codes in a dword
    const uint32_t x = a ^ b;
    return __builtin_popcount((x | (x >> 1)) & 0x55555555u);
}

__device__ __forceinline__ uint64_t block_sum_to(uint64_t v, unsigned long long* dst) {
    // wave reduce -> LDS -> one no-return atomic per workgroup
    __shared__ unsigned long long part[kRedBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)v, off, 64), hi = __shfl_down((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) s += part[w];
        if (s) (void)__hip_atomic_fetch_add(dst, (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return s;
    }
    return 0;
}

// Where the 16-B vector v of workgroup b's tile lives.  Linear: tile b, vector v.  XCD-interleaved
// (XI): groups of X workgroups (one per XCD, block b runs on XCD b % X, X = 2^xs from chip_info()) share a super-tile of X tiles
// and take its 4-KiB pages round-robin, so that at any time the eight XCDs read eight CONSECUTIVE
// pages and each one whole pages -- the access pattern the codec kernels get from their small tiles
// (codec2_kernels.hpp), here for 32-64 KiB reduction tiles.  Returns the byte offset from `base` and
// sets `base_off` to the start of the descriptor window.
template <uint32_t TILE, bool XI>
__device__ __forceinline__ uint32_t vec_offset(uint64_t b, uint64_t n_tiles, uint32_t v, uint32_t xs, uint64_t& win_base, uint32_t& win_bytes) {
    if constexpr (XI) {
        const uint64_t g = b >> xs;
        if (((g + 1) << xs) <= n_tiles) {
            win_base = (g << xs) * (uint64_t)TILE;
            win_bytes = TILE << xs;
            const uint32_t x = (uint32_t)b & ((1u << xs) - 1u), page = v >> 8, off = v & 255u;
            return (((page << xs) + x) << 12) + (off << 4);
        }
    }
    win_base = b * (uint64_t)TILE;
    win_bytes = TILE;
    return v << 4;
}

// Hamming distance over whole 16-B vectors (64 nt each); tile = kRedBlock*U vectors per workgroup.
// block_sum_to's sibling: the workgroup's sum goes to partial[blockIdx.x], no atomic
__device__ __forceinline__ void block_sum_store(uint64_t v, unsigned long long* partial) {
    __shared__ unsigned long long part2[kRedBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)v, off, 64), hi = __shfl_down((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part2[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t s = 0;
        for (unsigned w = 0; w < blockDim.x / 64; ++w) s += part2[w];
        partial[blockIdx.x] = s;
    }
}

// second pass: a few workgroups add the n partial sums into *count (one atomic each; a single
// workgroup took 35-60 us for the 10^5 partials of a 2^34-nt call, 3-4 % of the whole call)
__global__ __launch_bounds__(kRedBlock) void sum_partials(const unsigned long long* __restrict__ partial, uint64_t n,
                                                          unsigned long long* __restrict__ count) {
    uint64_t s = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)kRedBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kRedBlock) s += partial[i];
    block_sum_to(s, count);
}
inline unsigned sum_partials_grid(uint64_t n) {
    const uint64_t g = (n + 2 * kRedBlock - 1) / (2 * kRedBlock);  // >= 2 partials per thread
    return (unsigned)(g < 1 ? 1 : g > 128 ? 128 : g);
}

template <int U, bool XI>
__global__ __launch_bounds__(kRedBlock) void hamming_tiles(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                           uint64_t n_tiles, unsigned long long* __restrict__ partial, uint32_t xs) {
    constexpr uint32_t TILE = kRedBlock * U * 16;
    uint64_t wb;
    uint32_t wn, off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) off[u] = vec_offset<TILE, XI>(blockIdx.x, n_tiles, u * kRedBlock + threadIdx.x, xs, wb, wn);
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(a + wb, wn), rb = rsrc_of(b + wb, wn);
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        va[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off[u], 0, kNT));
        vb[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, off[u], 0, kNT));
    }
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
        c += diff_codes32(va[u].x, vb[u].x) + diff_codes32(va[u].y, vb[u].y) + diff_codes32(va[u].z, vb[u].z) + diff_codes32(va[u].w, vb[u].w);
    block_sum_store(c, partial);
}

// ---- persistent reductions (the shipped form since round 2) ------------------------------------------------
// A reduction has no stores, so a wave can stay resident, walk its share of the buffer with a deep software
// pipeline and finish with ONE atomic: wave g of G takes the contiguous RUN-KiB pieces g, g+G, g+2G, ... (chip-wide
// the waves read one compact, advancing window), the RUN (x2 streams) 1-KiB loads of the next piece are in flight
// while the current piece is counted.  bench/tune_lab14/15.hip: a few such waves per CU read at 7.2-7.4 TB/s, as
// fast as the one-shot tiles -- and a call is one launch with ~10^3 atomics instead of tiles + a stream-ordered
// scratch array (hipMallocAsync) + a second kernel + a free, which cost the API call 60-100 us on top of a 1.2 ms
// kernel (hamming 0.82-0.86 -> 0.90-0.92 of the roofline at the entry point).  No allocation: capturable in a graph.
// (wave_sum_to: codec2_kernels.hpp -- the checked encoders end in the same one-atomic-per-wave)

constexpr int kHammingRunKiB = 8, kHammingWavesPerCU = 2;     // bench/tune_lab15.hip
constexpr int kValidateRunKiB = 16, kValidateWavesPerCU = 4;
// the grid is min(n_runs, compute units x waves per CU); the CU count comes from chip_info() (codec2_launch.hpp)

template <int RUN>
__global__ __launch_bounds__(64) void hamming_persist(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n_runs,
                                                      unsigned long long* __restrict__ count) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    if (g >= n_runs) return;
    u32x4 va[RUN], vb[RUN];
    {
        const __amdgpu_buffer_rsrc_t ra = rsrc_of(a + g * (RUN * 1024ull), RUN * 1024), rb = rsrc_of(b + g * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) {
            va[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (d * 64 + lane) * 16, 0, kNT));
            vb[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (d * 64 + lane) * 16, 0, kNT));
        }
    }
    uint64_t total = 0;
    for (uint64_t t = g; t < n_runs; t += G) {
        const uint64_t nx = t + G < n_runs ? t + G : t;  // the last iteration re-reads its own piece; it is counted once
        const __amdgpu_buffer_rsrc_t ra = rsrc_of(a + nx * (RUN * 1024ull), RUN * 1024), rb = rsrc_of(b + nx * (RUN * 1024ull), RUN * 1024);
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < RUN; ++d) {
            c += diff_codes32(va[d].x, vb[d].x) + diff_codes32(va[d].y, vb[d].y) + diff_codes32(va[d].z, vb[d].z) + diff_codes32(va[d].w, vb[d].w);
            va[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (d * 64 + lane) * 16, 0, kNT));
            vb[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (d * 64 + lane) * 16, 0, kNT));
        }
        total += c;
    }
    wave_sum_to(total, count);
}

#ifdef CNT_LAB_VARIANTS
// Lab build, tuning key "hamming_order": other ways to issue the two streams' loads of a piece.  The shipped kernel above
// interleaves a[d], b[d]; ORDER 1 = all of a's, then all of b's; ORDER 2 = SKEWED: stream a runs one piece ahead of stream
// b, so the loads a wave has in flight aim at different offsets of the two operands (a[t + 2G] with b[t + G]).  Hamming
// is the one kernel whose rate moves with WHERE its operands lie (0.81-0.93 from box to box,
// profiles/r03_hamming_placement.jsonl): two streams read at equal offsets.
template <int RUN, int ORDER>
__global__ __launch_bounds__(64) void hamming_persist_lab(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n_runs,
                                                          unsigned long long* __restrict__ count) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    if (g >= n_runs) return;
    auto load = [&](const uint8_t* base, uint64_t piece, u32x4(&v)[RUN]) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(base + piece * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) v[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
    };
    auto diff = [&](const u32x4(&x)[RUN], const u32x4(&y)[RUN]) {
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < RUN; ++d) c += diff_codes32(x[d].x, y[d].x) + diff_codes32(x[d].y, y[d].y) + diff_codes32(x[d].z, y[d].z) + diff_codes32(x[d].w, y[d].w);
        return c;
    };
    const uint64_t last = g + (n_runs - 1 - g) / G * G;  // this wave's last piece; pieces past it are re-reads of it
    uint64_t total = 0;
    if constexpr (ORDER == 2) {
        u32x4 va[RUN], va2[RUN], vb[RUN];
        load(a, g, va);
        load(a, g + G <= last ? g + G : last, va2);
        load(b, g, vb);
        for (uint64_t t = g; t < n_runs; t += G) {
            total += diff(va, vb);
#pragma unroll
            for (int d = 0; d < RUN; ++d) va[d] = va2[d];
            load(a, t + 2 * G <= last ? t + 2 * G : last, va2);
            load(b, t + G <= last ? t + G : last, vb);
        }
    } else {
        u32x4 va[RUN], vb[RUN];
        load(a, g, va);
        load(b, g, vb);
        for (uint64_t t = g; t < n_runs; t += G) {
            total += diff(va, vb);
            load(a, t + G <= last ? t + G : last, va);
            load(b, t + G <= last ? t + G : last, vb);
        }
    }
    wave_sum_to(total, count);
}
#endif

// generic / tail: one thread per word from first_word, last word masked to `len`
__global__ __launch_bounds__(kRedBlock) void hamming_generic(const uint64_t* __restrict__ a, const uint64_t* __restrict__ b,
                                                             uint64_t len, uint64_t first_word, uint64_t n_words,
                                                             unsigned long long* __restrict__ count) {
    uint64_t c = 0;
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kRedBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kRedBlock) {
        uint64_t x = a[w] ^ b[w];
        const uint64_t rem = len - (w << 5);
        if (rem < 32) x &= (1ull << (2 * rem)) - 1;
        c += __builtin_popcountll((x | (x >> 1)) & 0x5555555555555555ull);
    }
    block_sum_to(c, count);
}

// complement: one thread per 16-B vector (whole vectors), stream shape of the codec kernels
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void complement_tiles(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE, TILE), rout = rsrc_of(out + t * TILE, TILE);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, kNT));
#pragma unroll
    for (int u = 0; u < U; ++u)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v[u] ^ 0xAAAAAAAAu), rout, (u * BLOCK + threadIdx.x) * 16, 0, kSC0 | kSC1 | kNT);
}

__global__ __launch_bounds__(kBlock) void complement_generic(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint64_t len,
                                                             uint64_t first_word, uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock) {
        uint64_t x = in[w] ^ 0xAAAAAAAAAAAAAAAAull;
        const uint64_t rem = len - (w << 5);
        if (rem < 32) x &= (1ull << (2 * rem)) - 1;  // unused high bits stay zero, like every encoder's output
        out[w] = x;
    }
}

// reverse the order of the 32 two-bit codes of a word
__device__ __forceinline__ uint64_t reverse_codes64(uint64_t x) {
    x = __brevll(x);  // reverses bits: also swaps the two bits inside each code ...
    return ((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1);  // ... swap them back
}

// reverse complement: output word w holds nt 32w..32w+31 = complement of input nt len-1-32w-k.
// One thread per output word; each reads a 64-bit window that straddles two input words.
__global__ __launch_bounds__(kBlock) void reverse_complement_words(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                   uint64_t len, uint64_t first_word, uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock) {
        // input nts p_lo .. p_lo+31 with p_lo = len - 32 - 32w (may be negative for the last output word)
        const int64_t p_lo = (int64_t)len - 32 - (int64_t)(w << 5);
        uint64_t window;
        if (p_lo >= 0) {
            const uint64_t iw = (uint64_t)p_lo >> 5;
            const unsigned sh = 2u * ((unsigned)p_lo & 31u);
            window = in[iw] >> sh;
            if (sh) window |= in[iw + 1] << (64 - sh);  // iw+1 <= (len-1)>>5 whenever sh != 0
        } else {
            window = in[0] << (2u * (unsigned)(-p_lo));  // the first -p_lo codes of the window do not exist
        }
        uint64_t x = reverse_codes64(window) ^ 0xAAAAAAAAAAAAAAAAull;
        const uint64_t rem = len - (w << 5);
        if (rem < 32) x &= (1ull << (2 * rem)) - 1;
        out[w] = x;
    }
}

// Tile form of the same: one workgroup = BLOCK*2 consecutive OUTPUT words, two per lane (one 16-B
// store, 4 KiB per workgroup).  Output words w0+2i, w0+2i+1 are the bit windows at input nt
// P - 64i and P - 64i - 32 with P = len - 32 - 32*w0, so the window phase (P & 31) and the word
// J = P >> 5 are tile-uniform and lane i needs input words J-2i-1, J-2i (one 16-B load at an
// 8-B-aligned address) and, when the phase is not 0, J-2i+1 (an 8-B load; with phase 0 the lane aims
// outside the descriptor and nothing is fetched).  The tile reads the input backwards, but each
// wave-instruction still covers one contiguous 1 KiB span.  Only whole tiles whose windows lie
// entirely inside the input: the launcher leaves the last len % (64*BLOCK) nt to the word kernel.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void reverse_complement_tiles(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                                   uint64_t len, uint64_t n_tiles) {
    constexpr uint32_t TILE_W = BLOCK * 2;
    const uint64_t w0 = (uint64_t)blockIdx.x * TILE_W;
    const uint64_t P = len - 32 - (w0 << 5);
    const uint64_t J = P >> 5;
    const uint32_t sh = 2u * ((uint32_t)P & 31u);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (J + 1 - TILE_W) * 8, (TILE_W + 2) * 8);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + w0 * 8, TILE_W * 8);
    const uint32_t i = threadIdx.x;
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    const u32x4 q = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (TILE_W - 2 - 2 * i) * 8, 0, kNT));
    const vu2 e = __builtin_amdgcn_raw_buffer_load_b64(rin, sh ? (TILE_W - 2 * i) * 8 : 0xFFFFFFF0u, 0, kNT);
    const uint64_t wm = ((uint64_t)q.y << 32) | q.x;   // input word j-1
    const uint64_t wj = ((uint64_t)q.w << 32) | q.z;   // input word j
    const uint64_t wp = ((uint64_t)e[1] << 32) | e[0];  // input word j+1 (0 when sh == 0)
    // branch-free: (x << 1) << (63 - sh) is x << (64 - sh) for sh = 2..62 and 0 for sh = 0.  With `if (sh)` around the two ORs
    // the compiler sank the second load into the branch, BEHIND the first load's s_waitcnt: two dependent trips to memory per
    // tile whenever len is not a multiple of 32 (tests/test_isa_digest.py now counts the loads in flight at the first wait).
    const uint64_t win0 = (wj >> sh) | ((wp << 1) << (63 - sh)), win1 = (wm >> sh) | ((wj << 1) << (63 - sh));
    const uint64_t o0 = reverse_codes64(win0) ^ 0xAAAAAAAAAAAAAAAAull, o1 = reverse_codes64(win1) ^ 0xAAAAAAAAAAAAAAAAull;
    const u32x4 o = {(uint32_t)o0, (uint32_t)(o0 >> 32), (uint32_t)o1, (uint32_t)(o1 >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, i * 16, 0, kSC0 | kSC1 | kNT);
}

// validity: count bytes outside the alphabet.  SWAR: zero byte in (x & 0xDF) ^ expect  <=> valid letter.
template <bool ALLOW_N>
__device__ __forceinline__ uint32_t invalid_bytes32(uint32_t x) {
    const uint32_t exp_lo = 0x43FF41FFu;                             // k=0:FF 1:'A' 2:FF 3:'C'
    const uint32_t exp_hi = ALLOW_N ? 0x474E5554u : 0x47FF5554u;     // k=4:'T' 5:'U' 6:'N'|FF 7:'G'
    const uint32_t expect = __builtin_amdgcn_perm(exp_hi, exp_lo, x & 0x07070707u);
    const uint32_t z = (x & 0xDFDFDFDFu) ^ expect;
    const uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
    return __builtin_popcount(nz);
}

template <int U, bool ALLOW_N, bool XI>
__global__ __launch_bounds__(kRedBlock) void validate_tiles(const uint8_t* __restrict__ n, uint64_t n_tiles,
                                                            unsigned long long* __restrict__ partial, uint32_t xs) {
    constexpr uint32_t TILE = kRedBlock * U * 16;
    uint64_t wb;
    uint32_t wn, off[U];
#pragma unroll
    for (int u = 0; u < U; ++u) off[u] = vec_offset<TILE, XI>(blockIdx.x, n_tiles, u * kRedBlock + threadIdx.x, xs, wb, wn);
    const __amdgpu_buffer_rsrc_t rn = rsrc_of(n + wb, wn);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rn, off[u], 0, kNT));
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < U; ++u)
        c += invalid_bytes32<ALLOW_N>(v[u].x) + invalid_bytes32<ALLOW_N>(v[u].y) + invalid_bytes32<ALLOW_N>(v[u].z) + invalid_bytes32<ALLOW_N>(v[u].w);
    block_sum_store(c, partial);
}

template <int RUN, bool ALLOW_N>
__global__ __launch_bounds__(64) void validate_persist(const uint8_t* __restrict__ n, uint64_t n_runs, unsigned long long* __restrict__ count) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    if (g >= n_runs) return;
    u32x4 v[RUN];
    {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(n + g * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) v[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
    }
    uint64_t total = 0;
    for (uint64_t t = g; t < n_runs; t += G) {
        const uint64_t nx = t + G < n_runs ? t + G : t;
        const __amdgpu_buffer_rsrc_t r = rsrc_of(n + nx * (RUN * 1024ull), RUN * 1024);
        uint32_t c = 0;
#pragma unroll
        for (int d = 0; d < RUN; ++d) {
            c += invalid_bytes32<ALLOW_N>(v[d].x) + invalid_bytes32<ALLOW_N>(v[d].y) + invalid_bytes32<ALLOW_N>(v[d].z) + invalid_bytes32<ALLOW_N>(v[d].w);
            v[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
        }
        total += c;
    }
    wave_sum_to(total, count);
}

template <bool ALLOW_N>
__global__ __launch_bounds__(kRedBlock) void validate_generic(const uint8_t* __restrict__ n, uint64_t first, uint64_t n_len,
                                                              unsigned long long* __restrict__ count) {
    uint64_t c = 0;
    for (uint64_t i = first + blockIdx.x * (uint64_t)kRedBlock + threadIdx.x; i < n_len; i += (uint64_t)gridDim.x * kRedBlock)
        c += invalid_bytes32<ALLOW_N>(0x41414100u | n[i]);  // pad the other three lanes with 'A' (valid)
    block_sum_to(c, count);
}

}  // namespace cnt
