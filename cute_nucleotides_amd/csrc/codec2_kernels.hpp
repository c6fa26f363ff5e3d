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
// dword against ~60 available per dword at full bandwidth), so everything here
// is about the memory system: 16 B per lane on the wide side, several
// independent loads in flight per lane, 64-bit tile indexing, optional
// non-temporal hints, and three ways of shaping the narrow side (kernel
// variants below, selected by measurement -- see DESIGN.md):
//
//   DIRECT   wide side 16 B/lane coalesced, narrow side 4 B/lane coalesced.
//   LDS      both sides 16 B/lane coalesced; a wave transposes its own
//            1 KiB x U tile through a private LDS slab (no block barrier).
//   LANE     each lane owns 64 consecutive nt: 4 x 16 B on the wide side
//            (lane-strided), one 16 B access on the narrow side, no LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cnt {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBlock = 256;  // 4 waves: one per SIMD
constexpr int kWave = 64;

// ---------------------------------------------------------------------------
// memory helpers
// ---------------------------------------------------------------------------
template <bool NT, typename T>
__device__ __forceinline__ T ld(const T* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT, typename T>
__device__ __forceinline__ void st(T* p, T v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Order a wave's LDS writes before its own later LDS reads of other lanes'
// data.  LDS instructions of one wave execute in issue order, so no s_barrier
// is needed -- only a compiler fence so the accesses are not reordered.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------
// encode arithmetic: 4 ASCII bytes (one dword) -> one packed byte
// ---------------------------------------------------------------------------
// Bits 1..2 of each byte are the code (n_to_bits.rs:85 mask 0x06).  With
// y = x & 0x06060606 the four codes sit at bits {1,2},{9,10},{17,18},{25,26};
// OR-ing y, y<<6, y<<12, y<<18 lines them up at bits 19..26 (checked
// exhaustively in tests/test_bit_tricks.py); two v_lshl_or_b32 build it.
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
// decode arithmetic: one packed byte -> 4 ASCII bytes (one dword)
// ---------------------------------------------------------------------------
// b | b<<6 | b<<12 | b<<18 puts code k at bits 8k..8k+1 (the spread the
// reference does with a carry-less multiply, n_to_bits.rs:357-365,381-384);
// v_perm_b32 is then a 4-entry byte LUT: code -> "ACTG" (n_to_bits.rs:23-30).
__device__ __forceinline__ uint32_t dec1(uint32_t b /* 0..255 */) {
    uint32_t t = (b << 6) | b;
    uint32_t sel = ((t << 12) | t) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, 0x47544341u /* 'A','C','T','G' = bytes 0..3 */, sel);
}

__device__ __forceinline__ u32x4 dec4(uint32_t x) {  // 4 packed bytes -> 16 ASCII bytes
    u32x4 r;
    r.x = dec1(x & 0xFFu);
    r.y = dec1(__builtin_amdgcn_ubfe(x, 8, 8));
    r.z = dec1(__builtin_amdgcn_ubfe(x, 16, 8));
    r.w = dec1(x >> 24);
    return r;
}

// ===========================================================================
// ENCODE kernels.  `in` = ASCII as 16-B vectors, tiles of kBlock*U vectors
// (kBlock*U*16 nt); only whole tiles are handled here, the remainder goes to
// n_to_bits_generic.  Tile index is 64-bit (2^36 nt = 2^32 lanes' worth).
// ===========================================================================

// DIRECT: dwordx4 load -> dword store, both coalesced.
template <int U, bool LNT, bool SNT, bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_direct(const u32x4* __restrict__ in, uint32_t* __restrict__ out,
                                                           uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LNT>(in + base + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) st<SNT>(out + base + u * kBlock, enc16<STRICT>(v[u]));
    }
}

// LDS: each wave loads U x 1 KiB coalesced, packs to U dwords per lane, writes
// them to its private LDS slab in output order (ds_write_b32, conflict-free),
// reads back 16 B per lane (ds_read_b128) and stores U/4 coalesced dwordx4.
template <int U, bool LNT, bool SNT, bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_lds(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                        uint64_t n_tiles) {
    static_assert(U % 4 == 0, "U must be a multiple of 4");
    __shared__ __attribute__((aligned(16))) uint32_t slab[kBlock / kWave][U * kWave];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* my = slab[wave];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t chunk = t * (kBlock / kWave) + wave;  // one wave-chunk = U KiB of ASCII
        const u32x4* src = in + chunk * (uint64_t)(U * kWave) + lane;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<LNT>(src + u * kWave);
#pragma unroll
        for (int u = 0; u < U; ++u) my[u * kWave + lane] = enc16<STRICT>(v[u]);
        wave_lds_fence();
        u32x4* dst = out + chunk * (uint64_t)(U * kWave / 4) + lane;
#pragma unroll
        for (int j = 0; j < U / 4; ++j) {
            u32x4 q = *reinterpret_cast<const u32x4*>(my + (j * kWave + lane) * 4);
            st<SNT>(dst + j * kWave, q);
        }
        wave_lds_fence();  // WAR: next tile's writes vs this tile's reads
    }
}

// LANE: each lane owns 64 consecutive nt (4 x 16 B loads at a 64-B lane
// stride) and stores one 16-B vector; no LDS.  R = such groups per lane.
template <int R, bool LNT, bool SNT, bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_lane(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                         uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * R) + threadIdx.x;  // in output-vector units
        u32x4 v[R][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) v[r][k] = ld<LNT>(in + (base + r * kBlock) * 4 + k);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u32x4 q;
            q.x = enc16<STRICT>(v[r][0]);
            q.y = enc16<STRICT>(v[r][1]);
            q.z = enc16<STRICT>(v[r][2]);
            q.w = enc16<STRICT>(v[r][3]);
            st<SNT>(out + base + r * kBlock, q);
        }
    }
}

// Generic / tail: one thread per output word, byte loads, any alignment, any
// length; zero-pads the last word (n_to_bits.rs:35).  first_word offsets both
// the input (32*first_word) and the output.
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits_generic(const uint8_t* __restrict__ n, uint64_t n_len,
                                                            uint64_t* __restrict__ out, uint64_t first_word,
                                                            uint64_t n_words) {
    for (uint64_t w = first_word + blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words;
         w += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i0 = w << 5;
        uint64_t acc = 0;
        const int m = (n_len - i0) < 32 ? (int)(n_len - i0) : 32;
        for (int k = 0; k < m; k += 4) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < m) x |= (uint32_t)n[i0 + k + j] << (8 * j);
            // a byte that was never loaded is 0 -> code 0 in both modes
            acc |= (uint64_t)__builtin_amdgcn_ubfe(enc4<STRICT>(x), 19, 8) << (2 * k);
        }
        out[w] = acc;
    }
}

// ===========================================================================
// DECODE kernels.  `out` = ASCII as 16-B vectors; whole tiles only.
// ===========================================================================

// DIRECT: dword load -> dwordx4 store, both coalesced.
template <int U, bool LNT, bool SNT>
__global__ __launch_bounds__(kBlock) void bits_to_n_direct(const uint32_t* __restrict__ in, u32x4* __restrict__ out,
                                                           uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * U) + threadIdx.x;
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld<LNT>(in + base + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) st<SNT>(out + base + u * kBlock, dec4(x[u]));
    }
}

// LDS: each wave loads V x 1 KiB of packed words as dwordx4 (coalesced),
// parks them in its LDS slab (ds_write_b128), re-reads dword (j*64+lane)
// (ds_read_b32, conflict-free) and stores 4V coalesced dwordx4 of ASCII.
template <int V, bool LNT, bool SNT>
__global__ __launch_bounds__(kBlock) void bits_to_n_lds(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                        uint64_t n_tiles) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[kBlock / kWave][V * kWave * 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t* my = slab[wave];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t chunk = t * (kBlock / kWave) + wave;  // V KiB packed -> 4V KiB ASCII
        const u32x4* src = in + chunk * (uint64_t)(V * kWave) + lane;
        u32x4 q[V];
#pragma unroll
        for (int v = 0; v < V; ++v) q[v] = ld<LNT>(src + v * kWave);
#pragma unroll
        for (int v = 0; v < V; ++v) *reinterpret_cast<u32x4*>(my + (v * kWave + lane) * 4) = q[v];
        wave_lds_fence();
        u32x4* dst = out + chunk * (uint64_t)(V * kWave * 4) + lane;
#pragma unroll
        for (int j = 0; j < 4 * V; ++j) st<SNT>(dst + j * kWave, dec4(my[j * kWave + lane]));
        wave_lds_fence();
    }
}

// LANE: each lane loads one 16-B vector of packed words and stores the 64 nt
// it expands to as 4 x 16 B at a 64-B lane stride; no LDS.
template <int R, bool LNT, bool SNT>
__global__ __launch_bounds__(kBlock) void bits_to_n_lane(const u32x4* __restrict__ in, u32x4* __restrict__ out,
                                                         uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * R) + threadIdx.x;  // in input-vector units
        u32x4 q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) q[r] = ld<LNT>(in + base + r * kBlock);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u32x4* dst = out + (base + r * kBlock) * 4;
            st<SNT>(dst + 0, dec4(q[r].x));
            st<SNT>(dst + 1, dec4(q[r].y));
            st<SNT>(dst + 2, dec4(q[r].z));
            st<SNT>(dst + 3, dec4(q[r].w));
        }
    }
}

// Generic / tail: one thread per packed word, writes min(32, len - 32w) bytes
// with byte stores; any alignment.  Bits beyond `len` are ignored
// (n_to_bits.rs:60-65 stops at len).
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
