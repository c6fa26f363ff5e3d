// util_kernels.hpp -- device-side synthetic input generator, checksum and
// compare, so multi-GiB buffers can be produced and verified without moving
// them over PCIe.  Definitions match oracle/cnt_oracle.c bit for bit (the
// oracle regenerates / re-checksums any chunk on the host).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "codec2_kernels.hpp"

namespace cnt {

constexpr uint64_t kGolden = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ uint64_t fmix64(uint64_t z) {  // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 16 two-bit draws -> 16 letters of "ACGT" (generator order, NOT the codec's)
__device__ __forceinline__ u32x4 acgt16(uint32_t r) {
    u32x4 o;
    uint32_t* po = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t b = (r >> (8 * j)) & 0xFFu;
        uint32_t t = (b << 6) | b;
        uint32_t sel = ((t << 12) | t) & 0x03030303u;
        po[j] = __builtin_amdgcn_perm(0u, 0x54474341u /* 'A','C','G','T' */, sel);
    }
    return o;
}

// One thread per 32-nt block.  Whole blocks with a 16-B aligned destination use
// two 16-B stores; the ragged last block / unaligned buffers use byte stores.
__global__ __launch_bounds__(kBlock) void fill_random_acgt(uint8_t* __restrict__ out, uint64_t first_block,
                                                           uint64_t n_len, uint64_t seed, int aligned16) {
    const uint64_t n_blocks = (n_len + 31) >> 5;
    for (uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_blocks; w += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r = fmix64(seed + (first_block + w + 1) * kGolden);
        const uint64_t i0 = w << 5;
        u32x4 lo = acgt16((uint32_t)r), hi = acgt16((uint32_t)(r >> 32));
        if (aligned16 && i0 + 32 <= n_len) {
            u32x4* dst = reinterpret_cast<u32x4*>(out + i0);
            dst[0] = lo;
            dst[1] = hi;
        } else {
            const uint32_t* pl = reinterpret_cast<const uint32_t*>(&lo);
            const uint32_t* ph = reinterpret_cast<const uint32_t*>(&hi);
            for (int k = 0; k < 32 && i0 + k < n_len; ++k) {
                uint32_t d = k < 16 ? pl[k >> 2] : ph[(k - 16) >> 2];
                out[i0 + k] = (uint8_t)(d >> (8 * (k & 3)));
            }
        }
    }
}

// One thread per 27-nt block, byte stores (utility path, not tuned).
__global__ __launch_bounds__(kBlock) void fill_random_acgtn(uint8_t* __restrict__ out, uint64_t first_block,
                                                            uint64_t n_len, uint64_t seed) {
    const uint64_t n_blocks = (n_len + 26) / 27;
    for (uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_blocks; w += (uint64_t)gridDim.x * kBlock) {
        const uint64_t r0 = fmix64(seed + (first_block + w + 1) * kGolden);
        const uint64_t r1 = fmix64(r0 + kGolden);
        const uint64_t r2 = fmix64(r1 + kGolden);
        const uint64_t i0 = w * 27;
        for (int k = 0; k < 27 && i0 + k < n_len; ++k) {
            const bool is_n = (((r1 >> (2 * k)) & 3) == 0) && (((r2 >> (2 * k)) & 3) == 0);
            const uint32_t c = (uint32_t)(r0 >> (2 * k)) & 3u;
            out[i0 + k] = is_n ? (uint8_t)'N' : (uint8_t)(0x54474341u >> (8 * c));
        }
    }
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t lo = __shfl_down((uint32_t)v, off, 64);
        uint32_t hi = __shfl_down((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void checksum_words(const uint64_t* __restrict__ w, uint64_t first_word,
                                                         uint64_t n_words, unsigned long long* __restrict__ sum) {
    uint64_t s = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * kBlock)
        s += fmix64(w[i] + (first_word + i + 1) * kGolden);
    s = wave_sum_u64(s);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (unsigned long long)s);
}

__global__ __launch_bounds__(kBlock) void count_mismatch(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                         uint64_t nbytes, int aligned16,
                                                         unsigned long long* __restrict__ count) {
    uint64_t c = 0;
    const uint64_t tid = blockIdx.x * (uint64_t)kBlock + threadIdx.x, nthr = (uint64_t)gridDim.x * kBlock;
    uint64_t done = 0;
    if (aligned16) {
        const uint64_t nvec = nbytes >> 4;
        const u32x4* va = reinterpret_cast<const u32x4*>(a);
        const u32x4* vb = reinterpret_cast<const u32x4*>(b);
        for (uint64_t i = tid; i < nvec; i += nthr) {
            u32x4 x = va[i] ^ vb[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t z = x[j];
                uint32_t nz = (((z & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | z) & 0x80808080u;
                c += __builtin_popcount(nz);
            }
        }
        done = nvec << 4;
    }
    for (uint64_t i = done + tid; i < nbytes; i += nthr) c += a[i] != b[i];
    c = wave_sum_u64(c);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace cnt
