// tune_lab7.hip -- shapes for the two-stream read+reduce kernel (Hamming distance) (bench only)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab7 bench/tune_lab7.hip
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

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t diff32(uint32_t a, uint32_t b) { const uint32_t x = a ^ b; return __builtin_popcount((x | (x >> 1)) & 0x55555555u); }

// MODE 0: one atomic per workgroup (LDS reduce); MODE 1: one atomic per wave; MODE 2: per-workgroup partial to out[blockIdx]
template <int BLOCK, int U, int MODE, int LAUX>
__global__ __launch_bounds__(BLOCK) void ham(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n_tiles,
                                             unsigned long long* __restrict__ count, unsigned long long* __restrict__ partial) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    __shared__ unsigned long long part[16];
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t ra = rsrc_of(a + t * TILE, TILE), rb = rsrc_of(b + t * TILE, TILE);
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        va[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
        vb[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    }
    uint32_t c = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) c += diff32(va[u].x, vb[u].x) + diff32(va[u].y, vb[u].y) + diff32(va[u].z, vb[u].z) + diff32(va[u].w, vb[u].w);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if constexpr (MODE == 1) {
        if ((threadIdx.x & 63) == 0) (void)__hip_atomic_fetch_add(count, (unsigned long long)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long s = 0;
            for (int w = 0; w < BLOCK / 64; ++w) s += part[w];
            if constexpr (MODE == 0) (void)__hip_atomic_fetch_add(count, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else partial[blockIdx.x] = s;
        }
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };
static uint8_t *d_a, *d_b; static unsigned long long *d_count, *d_partial; static uint64_t N;  // N = bytes of each packed input
static std::vector<Variant> vs;
template <int B, int U, int M, int L> void add() { char n[96]; snprintf(n, 96, "ham B=%-4d U=%d mode=%d ld=%-2d", B, U, M, L); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((ham<B, U, M, L>), dim3((unsigned)t), dim3(B), 0, s, d_a, d_b, t, d_count, d_partial); }, {}}); }

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 32, rounds = argc > 2 ? atoi(argv[2]) : 5, iters = argc > 3 ? atoi(argv[3]) : 3;
    N = 1ull << log2;  // 2^32 bytes = 4 GiB per input = 2^34 nt
    CK(hipMalloc(&d_a, N)); CK(hipMalloc(&d_b, N)); CK(hipMalloc(&d_count, 8)); CK(hipMalloc(&d_partial, 8ull << 24));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_a, (uint64_t)0, N, (uint64_t)1, 1);
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_b, (uint64_t)0, N, (uint64_t)2, 1);
    CK(hipStreamSynchronize(s));
    add<1024, 4, 0, 2>(); add<1024, 2, 0, 2>(); add<1024, 1, 0, 2>(); add<512, 2, 0, 2>(); add<512, 1, 0, 2>(); add<256, 4, 0, 2>(); add<256, 2, 0, 2>(); add<256, 1, 0, 2>();
    add<1024, 1, 2, 2>(); add<1024, 2, 2, 2>(); add<256, 2, 2, 2>(); add<256, 1, 2, 2>(); add<128, 2, 2, 2>(); add<64, 2, 2, 2>();
    add<256, 2, 1, 2>(); add<64, 4, 1, 2>(); add<1024, 1, 0, 0>(); add<1024, 2, 0, 3>(); add<1024, 1, 0, 18>();
    for (auto& v : vs) { v.launch(s); CK(hipGetLastError()); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-36s %8.4f ms (min %8.4f)  %7.1f GB/s  %7.1f Gnt/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 2.0 * N / v.ms[v.ms.size() / 2] / 1e6, 4.0 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
