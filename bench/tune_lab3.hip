// tune_lab3.hip -- no-arithmetic ceilings at the SHIPPED workgroup shapes (bench only): what
// do a read-only, a write-only, a 4:1 and a 1:4 stream reach when issued exactly like the
// codec kernels (buffer ops, small workgroups, same cache-policy bits)?  Gives the composite
// ceiling DESIGN.md compares the codec kernels with.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab3 bench/tune_lab3.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// read-only: U x 16-B loads per lane, xor-reduced, stored only if a (never true) condition holds
template <int BLOCK, int U, int LAUX>
__global__ __launch_bounds__(BLOCK) void k_read(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint64_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE, TILE);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[threadIdx.x] = acc;
}
// write-only: U x 16-B stores per lane
template <int BLOCK, int U, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_write(uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE, TILE);
    const u32x4 v = {(uint32_t)t, threadIdx.x, 3u, 4u};
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
}
// 4:1 -- the encode kernel's exact access shape without the arithmetic (dword = xor of the 4 loaded dwords)
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_r4w1(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w, rout, (u * BLOCK + threadIdx.x) * 4, 0, SAUX);
}
// 1:4 -- the decode kernel's shape without the arithmetic
template <int BLOCK, int U, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_r1w4(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    uint32_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + threadIdx.x) * 4, 0, LAUX);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u32x4 v = {x[u], x[u] + 1, x[u] + 2, x[u] + 3};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; double bytes; };
static uint8_t *d_a, *d_b;
static uint64_t N;
static std::vector<Variant> vs;

template <int B, int U, int L> void add_read() { char n[96]; snprintf(n, 96, "read   B=%-4d U=%d ld=%-2d", B, U, L); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((k_read<B, U, L>), dim3((unsigned)t), dim3(B), 0, s, d_a, d_b, t); }, {}, (double)N}); }
template <int B, int U, int S> void add_write() { char n[96]; snprintf(n, 96, "write  B=%-4d U=%d st=%-2d", B, U, S); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((k_write<B, U, S>), dim3((unsigned)t), dim3(B), 0, s, d_b, t); }, {}, (double)N}); }
template <int B, int U, int C, int L, int S> void add_r4w1() { char n[96]; snprintf(n, 96, "r4w1   B=%-4d U=%d C=%d ld=%-2d st=%-2d", B, U, C, L, S); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((k_r4w1<B, U, C, L, S>), dim3((unsigned)t), dim3(B), 0, s, d_a, d_b, t); }, {}, 1.25 * N}); }
template <int B, int U, int L, int S> void add_r1w4() { char n[96]; snprintf(n, 96, "r1w4   B=%-4d U=%d ld=%-2d st=%-2d", B, U, L, S); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((k_r1w4<B, U, L, S>), dim3((unsigned)t), dim3(B), 0, s, d_a, d_b, t); }, {}, 1.25 * N}); }

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 5, iters = argc > 3 ? atoi(argv[3]) : 2;
    N = 1ull << log2;
    CK(hipMalloc(&d_a, N)); CK(hipMalloc(&d_b, N));
    CK(hipMemset(d_a, 0x41, N)); CK(hipMemset(d_b, 0, N));
    hipStream_t s; CK(hipStreamCreate(&s));
    add_read<64, 1, 2>(); add_read<64, 2, 2>(); add_read<64, 4, 2>(); add_read<128, 2, 2>(); add_read<256, 1, 2>(); add_read<256, 2, 2>(); add_read<256, 4, 2>();
    add_read<64, 2, 0>(); add_read<64, 2, 3>(); add_read<64, 2, 18>(); add_read<256, 1, 0>(); add_read<512, 1, 2>(); add_read<1024, 1, 2>();
    add_write<64, 1, 19>(); add_write<64, 2, 19>(); add_write<128, 2, 19>(); add_write<256, 1, 19>(); add_write<256, 2, 19>(); add_write<256, 4, 19>();
    add_write<128, 2, 0>(); add_write<128, 2, 2>(); add_write<128, 2, 16>(); add_write<128, 2, 18>(); add_write<128, 2, 17>(); add_write<64, 4, 19>(); add_write<128, 1, 19>();
    add_r4w1<64, 2, 2, 2, 16>(); add_r4w1<64, 2, 1, 2, 16>(); add_r4w1<256, 1, 1, 2, 16>(); add_r4w1<256, 4, 1, 2, 2>(); add_r4w1<256, 4, 1, 0, 0>();
    add_r1w4<128, 2, 0, 19>(); add_r1w4<256, 2, 0, 19>(); add_r1w4<256, 4, 2, 2>(); add_r1w4<256, 4, 0, 0>();
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
    for (auto& v : vs) {
        std::sort(v.ms.begin(), v.ms.end());
        printf("%-40s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], v.bytes / v.ms[v.ms.size() / 2] / 1e6);
    }
    return 0;
}
