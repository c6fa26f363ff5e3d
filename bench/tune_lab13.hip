// tune_lab13.hip -- is the 4 : 1 ceiling a property of the KERNEL (each wave must read before it writes) or of the
// MEMORY SYSTEM?  Run a pure-read kernel over 16 GiB and a pure-write kernel over 4 GiB CONCURRENTLY on two
// streams -- no data dependence between any read and any write, the two kernels share the chip however the
// dispatcher interleaves them -- and compare the aggregate with the arithmetic-free 4 : 1 kernel of encode's
// shape and with the sum of the two alone.  Bench only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab13 bench/tune_lab13.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_read(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE = BLOCK * U * 16;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (uint64_t)blockIdx.x * TILE, TILE);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, kNT));
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[threadIdx.x] = acc;
    if (n_tiles == ~0ull) pad[threadIdx.x] = acc.x;
}
template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void k_write(uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE = BLOCK * U * 16;
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + (uint64_t)blockIdx.x * TILE, TILE);
    const u32x4 v = {(uint32_t)blockIdx.x, threadIdx.x, 3u, 4u};
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, kSC0 | kSC1 | kNT);
    if (n_tiles == ~0ull) pad[threadIdx.x] = v.x;
}
template <int BLOCK, int U, int C>
__global__ __launch_bounds__(BLOCK) void k_r4w1(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, kNT));
    if (n_tiles == ~0ull) pad[threadIdx.x] = v[0].x;
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w, rout, (u * BLOCK + threadIdx.x) * 4, 0, kSC0 | kSC1 | kNT);
}

static uint32_t lds_cap(int cap) { return cap ? (163840u / cap) / 256u * 256u : 0u; }

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 7;
    const uint64_t N = 1ull << log2;
    uint8_t *d_in, *d_out, *d_sink;
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_out, N / 4)); CK(hipMalloc(&d_sink, 1 << 20));
    CK(hipMemset(d_in, 0x41, N)); CK(hipMemset(d_out, 0, N / 4));
    hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    struct Shape { const char* name; int rcap, wcap; };
    // read side: encode's own read shape (64 thr x 2 loads, 2 KiB per workgroup); write side: 64 thr x 2 stores of 16 B
    // (2 KiB per workgroup, a quarter as many workgroups) -- and the probes' best shapes for comparison
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> alone_r, alone_w, both, fused;
        for (int r = 0; r < rounds + 1; ++r) {
            auto launch_r = [&](hipStream_t s) {
                if (mode == 0) hipLaunchKernelGGL((k_read<64, 2>), dim3((unsigned)(N / 2048)), dim3(64), lds_cap(23), s, d_in, d_sink, N / 2048);
                else if (mode == 1) hipLaunchKernelGGL((k_read<1024, 1>), dim3((unsigned)(N / 16384)), dim3(1024), 0, s, d_in, d_sink, N / 16384);
                else hipLaunchKernelGGL((k_read<64, 2>), dim3((unsigned)(N / 2048)), dim3(64), lds_cap(18), s, d_in, d_sink, N / 2048);
            };
            auto launch_w = [&](hipStream_t s) {
                if (mode == 0) hipLaunchKernelGGL((k_write<64, 2>), dim3((unsigned)(N / 4 / 2048)), dim3(64), lds_cap(23), s, d_out, N / 4 / 2048);
                else if (mode == 1) hipLaunchKernelGGL((k_write<256, 1>), dim3((unsigned)(N / 4 / 4096)), dim3(256), 0, s, d_out, N / 4 / 4096);
                else hipLaunchKernelGGL((k_write<64, 2>), dim3((unsigned)(N / 4 / 2048)), dim3(64), lds_cap(5), s, d_out, N / 4 / 2048);
            };
            float ms;
            CK(hipEventRecord(e0, sa)); launch_r(sa); CK(hipEventRecord(e1, sa)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) alone_r.push_back(ms);
            CK(hipEventRecord(e0, sb)); launch_w(sb); CK(hipEventRecord(e1, sb)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (r) alone_w.push_back(ms);
            // concurrent: wall clock from before both launches to after both streams are idle
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            launch_r(sa); launch_w(sb);
            CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
            const double w = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r) both.push_back(w);
            CK(hipDeviceSynchronize());
            t0 = std::chrono::steady_clock::now();
            hipLaunchKernelGGL((k_r4w1<64, 2, 2>), dim3((unsigned)(N / 2048)), dim3(64), lds_cap(23), sa, d_in, d_out, N / 2048);
            CK(hipStreamSynchronize(sa));
            const double f = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (r) fused.push_back(f);
        }
        const double r_ms = med(alone_r), w_ms = med(alone_w), b_ms = med(both), f_ms = med(fused);
        printf("mode %d (%s)\n", mode, mode == 0 ? "read 64x2 cap 23 | write 64x2 cap 23" : mode == 1 ? "read 1024x1 | write 256x1 (the probes' best shapes)" : "read 64x2 cap 18 | write 64x2 cap 5");
        printf("  read alone   %.4f ms = %7.1f GB/s\n  write alone  %.4f ms = %7.1f GB/s (N/4 bytes)\n", r_ms, N / r_ms / 1e6, w_ms, N / 4 / w_ms / 1e6);
        printf("  read || write, two streams, wall  %.4f ms = %7.1f GB/s of 1.25 N   (sum of the two alone: %.4f ms = %7.1f GB/s)\n", b_ms, 1.25 * N / b_ms / 1e6,
               r_ms + w_ms, 1.25 * N / (r_ms + w_ms) / 1e6);
        printf("  4:1 kernel of encode's shape, wall %.4f ms = %7.1f GB/s\n", f_ms, 1.25 * N / f_ms / 1e6);
    }
    return 0;
}
