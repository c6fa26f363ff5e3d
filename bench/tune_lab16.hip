// tune_lab16.hip -- where does the 5-letter encoder (0.82-0.83 of the roofline) sit against its own access pattern?
// Three kernels in the shipped launch shape (one wave, 2 words per lane = 3456 B in / 1 KiB out, 15 wg/CU):
//   shipped      n_to_bits2_wave<1, 2, nt, sc1>
//   no-arith     same global loads, same LDS staging (write 16 B, fence, read 8 dwords per word), trivial arithmetic
//   no-LDS       same global loads and stores, no LDS at all (the loads are xor-ed straight into the stored words)
// Bench only.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab16 bench/tune_lab16.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/codec5_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/util_kernels.hpp"

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>  // 1 = LDS staging without arithmetic, 2 = no LDS
__global__ __launch_bounds__(64) void enc5_probe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t my[kWaveDwords5 * WPL + 4];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, TILE_BYTES);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    constexpr int NLD = (TILE_VECS + 63) / 64;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        v[i] = u32x4{0, 0, 0, 0};
        if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64))
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, kNT));
    }
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const uint32_t q = (27u * lane + (uint32_t)kWaveBytes5 * j) >> 2;
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int d = 0; d < 8; d += 2) { a ^= my[q + d]; b ^= my[q + d + 1]; }
            const vu2 w2 = {a, b};
            __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, kSC1);
        }
    } else {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) { a ^= v[i].x ^ v[i].z; b ^= v[i].y ^ v[i].w; }
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const vu2 w2 = {a + j, b};
            __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, kSC1);
        }
        if (n_tiles == ~0ull) my[lane] = a;
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };

int main(int argc, char** argv) {
    const int log2w = argc > 1 ? atoi(argv[1]) : 29, rounds = argc > 2 ? atoi(argv[2]) : 7;
    const uint64_t words = 1ull << log2w, N = 27 * words, tiles = words / 128;
    uint8_t *d_in, *d_out;
    CK(hipMalloc(&d_in, N + 256)); CK(hipMalloc(&d_out, words * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgtn, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED);
    CK(hipStreamSynchronize(s));
    const uint32_t lds15 = (163840u / 15) / 256 * 256 - 3584u;
    std::vector<Variant> vs;
    for (int rep = 0; rep < 2; ++rep) {
        vs.push_back({"5-letter encode shipped (wave tiles, LDS, 15 wg/CU)", [=](hipStream_t st) { hipLaunchKernelGGL((n_to_bits2_wave<1, 2, kNT, kSC1, false>), dim3((unsigned)tiles), dim3(64), lds15, st, d_in, d_out, tiles); }, {}});
        vs.push_back({"  same loads + LDS staging, no arithmetic", [=](hipStream_t st) { hipLaunchKernelGGL((enc5_probe<1>), dim3((unsigned)tiles), dim3(64), lds15, st, d_in, d_out, tiles); }, {}});
        vs.push_back({"  same loads and stores, no LDS", [=](hipStream_t st) { hipLaunchKernelGGL((enc5_probe<2>), dim3((unsigned)tiles), dim3(64), lds15, st, d_in, d_out, tiles); }, {}});
    }
    for (auto& v : vs) { v.launch(s); CK(hipGetLastError()); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s)); v.launch(s); v.launch(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / 2);
        }
    const double bytes = (double)N + 8.0 * words;
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-56s %8.4f ms (min %8.4f)  %7.1f GB/s  %.3f of 8 TB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], bytes / v.ms[v.ms.size() / 2] / 1e6, bytes / v.ms[v.ms.size() / 2] / 8e9); }
    return 0;
}
