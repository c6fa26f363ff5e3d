// tune_lab11.hip -- the fused round trip's own sweep: workgroup shape x XCD group
// size x residency cap x store policy per output stream.  round_trip_stream shipped with encode's tile
// shape (64 thr x 2 loads, XCD pairs) and cap 13 without ever being swept on its own; its traffic mix
// (1 B read : 1.25 B written per nt) is decode's, not encode's.  Bench only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab11 bench/tune_lab11.hip
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

// ORDER 0: per load u: packed store then ASCII store (shipped).  1: all packed stores first, then all ASCII
// stores.  2: all ASCII stores first.
template <int BLOCK, int U, int C, int LAUX, int SP, int SB, int ORDER>
__global__ __launch_bounds__(BLOCK) void rt11(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_PK = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rpk = rsrc_of(packed + t * TILE_PK, TILE_PK);
    const __amdgpu_buffer_rsrc_t rbk = rsrc_of(back + t * TILE_IN, TILE_IN);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
    uint32_t code[U];
#pragma unroll
    for (int u = 0; u < U; ++u) code[u] = enc16<false>(v[u]);
    if constexpr (ORDER == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            __builtin_amdgcn_raw_buffer_store_b32(code[u], rpk, (u * BLOCK + tid) * 4, 0, SP);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(code[u])), rbk, (u * BLOCK + tid) * 16, 0, SB);
        }
    } else {
        if constexpr (ORDER == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(code[u], rpk, (u * BLOCK + tid) * 4, 0, SP);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(code[u])), rbk, (u * BLOCK + tid) * 16, 0, SB);
        if constexpr (ORDER == 2) {
#pragma unroll
            for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(code[u], rpk, (u * BLOCK + tid) * 4, 0, SP);
        }
    }
}

// the fused kernel's access pattern without its arithmetic: packed dword = xor of the four input dwords, the "decoded"
// vector = the input vector (same 16 B in, 4 B + 16 B out per lane and load)
template <int BLOCK, int U, int C>
__global__ __launch_bounds__(BLOCK) void rt_noarith(const uint8_t* __restrict__ in, uint8_t* __restrict__ packed, uint8_t* __restrict__ back, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_PK = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rpk = rsrc_of(packed + t * TILE_PK, TILE_PK), rbk = rsrc_of(back + t * TILE_IN, TILE_IN);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, kNT));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        __builtin_amdgcn_raw_buffer_store_b32(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w, rpk, (u * BLOCK + tid) * 4, 0, kSC0 | kSC1 | kNT);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v[u]), rbk, (u * BLOCK + tid) * 16, 0, kSC0 | kSC1 | kNT);
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };
static uint8_t *d_in, *d_packed, *d_out;
static uint64_t N;
static std::vector<Variant> vs;

template <int B, int U, int C, int L, int SP, int SB, int ORDER = 0> void add(int cap) {
    char n[128]; snprintf(n, 128, "rt B=%-3d U=%d C=%d ld=%-2d stp=%-2d stb=%-2d ord=%d cap=%-2d", B, U, C, L, SP, SB, ORDER, cap); uint64_t t = N / (B * U * 16);
    size_t lds = cap ? (size_t)(163840 / cap) / 256 * 256 : 0;
    CK(hipFuncSetAttribute((const void*)rt11<B, U, C, L, SP, SB, ORDER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((rt11<B, U, C, L, SP, SB, ORDER>), dim3((unsigned)t), dim3(B), lds, s, d_in, d_packed, d_out, t); }, {}}); }

static uint64_t checksum(const void* p, uint64_t words, hipStream_t s) {
    static unsigned long long* d_sum = nullptr;
    if (!d_sum) CK(hipMalloc(&d_sum, 8));
    CK(hipMemsetAsync(d_sum, 0, 8, s));
    hipLaunchKernelGGL(checksum_words, dim3(4096), dim3(kBlock), 0, s, static_cast<const uint64_t*>(p), (uint64_t)0, words, d_sum);
    unsigned long long h = 0; CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); return h;
}

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 5, iters = argc > 3 ? atoi(argv[3]) : 2;
    const int set = argc > 4 ? atoi(argv[4]) : 0;
    N = 1ull << log2;
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_packed, N / 4)); CK(hipMalloc(&d_out, N));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    CK(hipStreamSynchronize(s));
    constexpr int A = kSC0 | kSC1 | kNT, W = kSC1, WN = kSC1 | kNT;
    add<64, 2, 2, kNT, A, A>(13);  // shipped
    if (set == 0) {
        for (int k : {9, 10, 11, 12, 13, 14, 15, 16, 18, 20, 23}) { add<64, 2, 1, kNT, A, A>(k); add<64, 2, 2, kNT, A, A>(k); add<64, 2, 4, kNT, A, A>(k); }
        for (int k : {5, 6, 7, 8, 9, 10}) { add<128, 2, 1, kNT, A, A>(k); add<128, 2, 2, kNT, A, A>(k); add<128, 2, 4, kNT, A, A>(k); }
        for (int k : {5, 6, 7, 8, 10}) { add<64, 4, 1, kNT, A, A>(k); add<64, 4, 2, kNT, A, A>(k); }
        for (int k : {3, 4}) { add<256, 2, 1, kNT, A, A>(k); add<256, 2, 2, kNT, A, A>(k); }
    } else if (set == 3) {  // the shipped fused shape against its own arithmetic-free stream, alternating
        for (int rep = 0; rep < 3; ++rep) {
            add<64, 4, 1, kNT, A, A>(8);
            const uint64_t t = N / 4096; const size_t lds = (size_t)(163840 / 8) / 256 * 256;
            vs.push_back({"rt NO ARITHMETIC B=64 U=4 C=1 cap=8 (checksum differs by design)", [t, lds](hipStream_t s) { hipLaunchKernelGGL((rt_noarith<64, 4, 1>), dim3((unsigned)t), dim3(64), lds, s, d_in, d_packed, d_out, t); }, {}});
        }
    } else if (set == 2) {  // around the winners of set 0: 4-KiB ASCII tiles per workgroup, no XCD grouping
        for (int k : {7, 8, 9, 10, 11}) { add<64, 4, 1, kNT, A, A>(k); add<128, 2, 1, kNT, A, A>(k); add<64, 4, 1, kNT, A, A, 1>(k); }
        for (int k : {4, 5, 6}) { add<64, 8, 1, kNT, A, A>(k); add<128, 4, 1, kNT, A, A>(k); add<256, 2, 1, kNT, A, A>(k); }
        for (int k : {8, 9}) { add<64, 4, 1, kNT, W, A>(k); add<64, 4, 1, kNT, A, WN>(k); add<64, 4, 1, kSC0 | kNT, A, A>(k); add<64, 4, 1, 0, A, A>(k); }
        for (int k : {12, 14, 16}) { add<64, 3, 1, kNT, A, A>(k); }
    } else {
        for (int k : {12, 13, 14}) {
            add<64, 2, 2, kNT, W, A>(k); add<64, 2, 2, kNT, A, WN>(k); add<64, 2, 2, kNT, W, WN>(k); add<64, 2, 2, 0, A, A>(k);
            add<64, 2, 2, kSC0 | kNT, A, A>(k); add<64, 2, 2, kNT, A, A, 1>(k); add<64, 2, 2, kNT, A, A, 2>(k);
            add<64, 2, 2, kNT, 0, A>(k); add<64, 2, 2, kNT, kNT, A>(k); add<64, 2, 2, kNT, A, kNT>(k);
        }
    }
    add<64, 2, 2, kNT, A, A>(13);
    uint64_t ref_p = 0, ref_b = 0; bool have = false;
    for (auto& v : vs) {
        CK(hipMemsetAsync(d_out, 0xFF, 1 << 20, s)); CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        const uint64_t p = checksum(d_packed, N / 32, s), b = checksum(d_out, N / 8, s);
        if (v.name.find("NO ARITHMETIC") != std::string::npos) continue;
        if (!have) { ref_p = p; ref_b = b; have = true; }
        if (p != ref_p || b != ref_b) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
    }
    { v0_again: ; }
    vs[0].launch(s);
    if (checksum(d_out, N / 8, s) != ref_b || checksum(d_in, N / 8, s) != ref_b) { fprintf(stderr, "decoded text != input\n"); return 2; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-52s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 2.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
