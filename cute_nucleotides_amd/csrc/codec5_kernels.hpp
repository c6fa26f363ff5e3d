// codec5_kernels.hpp -- gfx950 kernels for the 5-letter {A,C,G,T/U,N} codec.
//
// Replaces the loops of n_to_bits2_{lut,pext} (reference src/n_to_bits2.rs:37-74,
// :118-189) and bits_to_n2_{lut,pdep} (:78-107, :196-268): 3 nt -> a + 5b + 25c
// (7 bits), 9 triplets (27 nt) per u64, bit 63 always 0.
//
// 27-nt words do not line up with power-of-two vectors (64 B of ASCII = 2.37
// words), so a workgroup stages a tile of ASCII through LDS with coalesced
// 16-B global accesses and each lane then works on one whole word from LDS.
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
    uint32_t c = (v * 41u) >> 10;       // v/25 for v < 128 (checked in tests/test_bit_tricks.py)
    uint32_t r = v - c * 25u;
    uint32_t b = (r * 13u) >> 6;        // r/5 for r < 69
    uint32_t a = r - b * 5u;
    c = c > 4u ? 4u : c;
    return a | (b << 8) | (c << 16);
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
// Tiled kernels: a workgroup handles kWordsPerTile words = 27*kWordsPerTile nt.
// 27 * 256 = 6912 B = 432 x 16 B: whole 16-B vectors, so tiles start 16-B
// aligned whenever the buffer does.
// ---------------------------------------------------------------------------
constexpr int kWords5 = kBlock;                 // words per tile: one per lane
constexpr int kTileBytes5 = 27 * kWords5;       // 6912
constexpr int kTileVecs5 = kTileBytes5 / 16;    // 432
static_assert(kTileBytes5 % 16 == 0, "tile must be whole 16-B vectors");

// Encode: coalesced 16-B loads of the ASCII tile into LDS, then lane l reads
// its 27 bytes (7 unaligned-safe dword reads via byte-assembled funnel shifts),
// packs, and stores one u64 (8 B/lane, coalesced).
template <bool STRICT>
__global__ __launch_bounds__(kBlock) void n_to_bits2_tiled(const u32x4* __restrict__ in, uint64_t* __restrict__ out,
                                                           uint64_t n_tiles) {
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileBytes5 / 4 + 4];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const u32x4* src = in + t * (uint64_t)kTileVecs5;
        u32x4 a = src[threadIdx.x];
        u32x4 b;
        const bool second = threadIdx.x + kBlock < kTileVecs5;
        if (second) b = src[threadIdx.x + kBlock];
        __syncthreads();  // previous iteration's readers are done
        *reinterpret_cast<u32x4*>(tile + threadIdx.x * 4) = a;
        if (second) *reinterpret_cast<u32x4*>(tile + (threadIdx.x + kBlock) * 4) = b;
        __syncthreads();
        // lane's 27 bytes start at byte 27*l: dword index q = (27*l)>>2, byte phase s = (27*l)&3
        const uint32_t byte0 = 27u * threadIdx.x;
        const uint32_t q = byte0 >> 2, s8 = (byte0 & 3u) * 8u;
        uint32_t raw[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) raw[d] = tile[q + d];  // q+7 <= 1727+... within the +4 pad
        uint32_t c[7];
#pragma unroll
        for (int d = 0; d < 7; ++d) {
            // funnel shift right by s8 bits across raw[d], raw[d+1]
            uint32_t x = (uint32_t)((((uint64_t)raw[d + 1] << 32) | raw[d]) >> s8);
            if (d == 6) x &= 0x00FFFFFFu;  // bytes 24..26 only
            c[d] = code5<STRICT>(x);
        }
        out[t * (uint64_t)kWords5 + threadIdx.x] = pack27(c);
    }
}

// Decode: lane l expands word l into 27 letters written to LDS (byte stores
// assembled as dwords is awkward at a 27-B pitch, so write bytes via
// 7 funnel-merged dwords is avoided: each lane writes its 27 bytes with
// ds_write_b8), then the tile leaves with coalesced 16-B stores.
__global__ __launch_bounds__(kBlock) void bits_to_n2_tiled(const uint64_t* __restrict__ in, u32x4* __restrict__ out,
                                                           uint64_t n_tiles) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[kTileBytes5];
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t word = in[t * (uint64_t)kWords5 + threadIdx.x];
        __syncthreads();
        uint8_t* dst = tile + 27u * threadIdx.x;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            uint32_t l = letters5(digits3((uint32_t)(word >> (7 * k)) & 0x7Fu));
            dst[3 * k + 0] = (uint8_t)l;
            dst[3 * k + 1] = (uint8_t)(l >> 8);
            dst[3 * k + 2] = (uint8_t)(l >> 16);
        }
        __syncthreads();
        u32x4* o = out + t * (uint64_t)kTileVecs5;
        o[threadIdx.x] = *reinterpret_cast<const u32x4*>(tile + threadIdx.x * 16);
        if (threadIdx.x + kBlock < kTileVecs5)
            o[threadIdx.x + kBlock] = *reinterpret_cast<const u32x4*>(tile + (threadIdx.x + kBlock) * 16);
    }
}

}  // namespace cnt
