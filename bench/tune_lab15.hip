// tune_lab15.hip -- reductions (Hamming distance, validity count) as PERSISTENT waves: lab 14 showed that a few
// waves per CU with a deep software pipeline READ at 7.3 TB/s; a reduction has no stores, so a persistent wave can
// accumulate over its whole share of the buffer and finish with ONE atomic -- ~1000 atomics per call instead of a
// partial-sum array (hipMallocAsync), a second kernel and a free.  Compared with the shipped tiles + second pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab15 bench/tune_lab15.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/util_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/packed_ops_kernels.hpp"

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void wave_sum_to(uint64_t v, unsigned long long* dst) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = __shfl_down((uint32_t)v, off, 64), hi = __shfl_down((uint32_t)(v >> 32), off, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0 && v) (void)__hip_atomic_fetch_add(dst, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave g of G takes the contiguous RUN-KiB pieces g, g+G, ... of BOTH streams; the loads of the next piece are in
// flight while the current one is counted
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
        const uint64_t nx = t + G < n_runs ? t + G : t;  // the last iteration reloads its own piece (counted once)
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

template <int RUN>
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
            c += invalid_bytes32<false>(v[d].x) + invalid_bytes32<false>(v[d].y) + invalid_bytes32<false>(v[d].z) + invalid_bytes32<false>(v[d].w);
            v[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
        }
        total += c;
    }
    wave_sum_to(total, count);
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; double bytes; bool ham; };
static uint8_t *d_a, *d_b, *d_n; static unsigned long long *d_count, *d_partial; static uint64_t NB, NN;  // NB = bytes per packed stream, NN = ASCII bytes
static std::vector<Variant> vs;
template <int RUN> void add_h(int wpc) { char n[96]; snprintf(n, 96, "hamming  persistent run=%-2d KiB waves/CU=%-2d", RUN, wpc); const uint64_t t = NB / (RUN * 1024ull); const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((hamming_persist<RUN>), dim3(g), dim3(64), 0, s, d_a, d_b, t, d_count); }, {}, 2.0 * NB, true}); }
template <int RUN> void add_v(int wpc) { char n[96]; snprintf(n, 96, "validate persistent run=%-2d KiB waves/CU=%-2d", RUN, wpc); const uint64_t t = NN / (RUN * 1024ull); const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((validate_persist<RUN>), dim3(g), dim3(64), 0, s, d_n, t, d_count); }, {}, (double)NN, false}); }

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 5;
    NN = 1ull << log2; NB = NN / 4;
    CK(hipMalloc(&d_a, NB)); CK(hipMalloc(&d_b, NB)); CK(hipMalloc(&d_n, NN)); CK(hipMalloc(&d_count, 8)); CK(hipMalloc(&d_partial, 8 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_n, (uint64_t)0, NN, (uint64_t)0x5EED, 1);
    hipLaunchKernelGGL((n_to_bits_stream<256, 4, 1, 0, 0, false>), dim3((unsigned)(NN / 16384)), dim3(256), 0, s, d_n, d_a, NN / 16384);
    CK(hipMemsetAsync(d_b, 0x1B, NB, s));  // the other sequence: ACGT ACGT ... -> differs from random codes in 3/4 of the positions
    CK(hipStreamSynchronize(s));
    {   // shipped: tiles (XCD-interleaved pages) + second pass
        constexpr int U = 2; constexpr uint64_t TILE = (uint64_t)kRedBlock * U * 16; const uint64_t t = NB / TILE;
        vs.push_back({"hamming  shipped: tiles + second pass", [t](hipStream_t st) {
            hipLaunchKernelGGL((hamming_tiles<U, true>), dim3((unsigned)t), dim3(kRedBlock), 0, st, d_a, d_b, t, d_partial);
            hipLaunchKernelGGL(sum_partials, dim3(sum_partials_grid(t)), dim3(kRedBlock), 0, st, d_partial, t, d_count); }, {}, 2.0 * NB, true});
        constexpr int UV = 4; constexpr uint64_t TV = (uint64_t)kRedBlock * UV * 16; const uint64_t tv = NN / TV;
        vs.push_back({"validate shipped: tiles + second pass", [tv](hipStream_t st) {
            hipLaunchKernelGGL((validate_tiles<UV, false, true>), dim3((unsigned)tv), dim3(kRedBlock), 0, st, d_n, tv, d_partial);
            hipLaunchKernelGGL(sum_partials, dim3(sum_partials_grid(tv)), dim3(kRedBlock), 0, st, d_partial, tv, d_count); }, {}, (double)NN, false});
    }
    for (int w : {2, 3, 4, 6, 8}) { add_h<4>(w); add_h<8>(w); add_h<16>(w); }
    for (int w : {2, 3, 4, 6, 8}) { add_v<8>(w); add_v<16>(w); add_v<32>(w); }
    unsigned long long ref_h = 0, ref_v = 0; bool hh = false, hv = false;
    for (auto& v : vs) {
        CK(hipMemsetAsync(d_count, 0, 8, s)); v.launch(s); CK(hipGetLastError());
        unsigned long long c = 0; CK(hipMemcpyAsync(&c, d_count, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        if (v.ham) { if (!hh) { ref_h = c; hh = true; } if (c != ref_h) { fprintf(stderr, "MISMATCH %s: %llu vs %llu\n", v.name.c_str(), c, ref_h); return 2; } }
        else { if (!hv) { ref_v = c; hv = true; } if (c != ref_v) { fprintf(stderr, "MISMATCH %s: %llu vs %llu\n", v.name.c_str(), c, ref_v); return 2; } }
    }
    printf("hamming = %llu of %llu nt, invalid = %llu\n", ref_h, (unsigned long long)NN, ref_v);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipMemsetAsync(d_count, 0, 8, s));
            CK(hipEventRecord(e0, s)); v.launch(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms);
        }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-48s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], v.bytes / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
