// tune_lab10.hip -- decode shapes not covered by labs 2/5: lane->tile layouts (wave-contiguous
// pieces), 256x1 / 128x1 / 64x4 workgroups under residency caps, XCD maps.  Bench only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab10 bench/tune_lab10.hip
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

// LY 0: vector index u*BLOCK + tid (shipped).  LY 1: each wave owns U consecutive 1-KiB pieces.
template <int BLOCK, int U, int C, int LY, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void dec10(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    uint32_t idx[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idx[u] = LY == 0 ? u * BLOCK + tid : ((tid >> 6) * U + u) * 64 + (tid & 63);
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, idx[u] * 4, 0, LAUX);
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = x[0];
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x[u])), rout, idx[u] * 16, 0, SAUX);
}


template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void enc10(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * BLOCK + tid) * 4, 0, SAUX);
}

// which of the 8 chunks of a group the XCD slot (b % 8) takes: (slot + ROT) % 8 or slot ^ XM
template <int C, int ROT, int XM>
__device__ __forceinline__ uint64_t tile_rot(uint64_t b, uint64_t n_tiles) {
    constexpr uint64_t G = 8 * C;
    const uint64_t g = b / G;
    if ((g + 1) * G > n_tiles) return b;
    const uint64_t r = b % G;
    const uint64_t slot = ((r % 8 + ROT) % 8) ^ XM;
    return g * G + slot * C + (r / 8);
}
template <int ROT, int XM>
__global__ __launch_bounds__(64) void enc_rot(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    const uint64_t t = tile_rot<2, ROT, XM>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 2048, 2048);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 512, 512);
    const uint32_t tid = threadIdx.x;
    u32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * 64 + tid) * 16, 0, kNT));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * 64 + tid) * 4, 0, kSC0 | kSC1 | kNT);
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; bool is_enc = false; };
static uint8_t *d_in, *d_packed, *d_out;
static uint64_t N;
static std::vector<Variant> vs;
template <int B, int U, int C, int LY, int L, int S> void add(int cap) {
    char n[96]; snprintf(n, 96, "dec B=%-3d U=%d C=%d ly=%d ld=%-2d st=%-2d cap=%-2d", B, U, C, LY, L, S, cap); uint64_t t = N / (B * U * 16);
    size_t lds = cap ? (size_t)(163840 / cap) / 256 * 256 : 0;
    CK(hipFuncSetAttribute((const void*)dec10<B, U, C, LY, L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((dec10<B, U, C, LY, L, S>), dim3((unsigned)t), dim3(B), lds, s, d_packed, d_out, t); }, {}}); }

template <int B, int U, int C, int L, int S> void add_enc(int cap) {
    char n[96]; snprintf(n, 96, "enc B=%-3d U=%d C=%-2d ld=%-2d st=%-2d cap=%-2d", B, U, C, L, S, cap); uint64_t t = N / (B * U * 16);
    size_t lds = cap ? (size_t)(163840 / cap) / 256 * 256 : 0;
    CK(hipFuncSetAttribute((const void*)enc10<B, U, C, L, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((enc10<B, U, C, L, S>), dim3((unsigned)t), dim3(B), lds, s, d_in, d_packed, t); }, {}, true}); }

template <int ROT, int XM> void add_rot() {
    char n[96]; snprintf(n, 96, "enc B=64 U=2 C=2 rot=%d xor=%d cap=23", ROT, XM); uint64_t t = N / 2048;
    size_t lds = (size_t)(163840 / 23) / 256 * 256;
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((enc_rot<ROT, XM>), dim3((unsigned)t), dim3(64), lds, s, d_in, d_packed, t); }, {}, true}); }


// map2: one input page per XCD turn as shipped, but the four pages whose packed output shares one
// 4-KiB output page go to the SAME XCD in four consecutive turns (groups of 64 tiles = 32 pages)
template <int MODE>
__global__ __launch_bounds__(64) void enc_map2(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    uint64_t t = blockIdx.x;
    const uint64_t g = t / 64;
    if ((g + 1) * 64 <= n_tiles) {
        const uint32_t r = t % 64, turn = r / 16, slot = r % 8, half = (r % 16) / 8;
        const uint32_t page = MODE == 0 ? 4 * slot + turn : 4 * slot + ((turn + slot) & 3);
        t = g * 64 + 2 * page + half;
    }
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 2048, 2048);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 512, 512);
    const uint32_t tid = threadIdx.x;
    u32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * 64 + tid) * 16, 0, kNT));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * 64 + tid) * 4, 0, kSC0 | kSC1 | kNT);
}
template <int MODE> void add_map2(int cap) {
    char n[96]; snprintf(n, 96, "enc map2 mode=%d cap=%d", MODE, cap); uint64_t t = N / 2048;
    size_t lds = (size_t)(163840 / cap) / 256 * 256;
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((enc_map2<MODE>), dim3((unsigned)t), dim3(64), lds, s, d_in, d_packed, t); }, {}, true}); }


// residency limited per SIMD through the occupancy attribute instead of per CU through dummy LDS
template <int WPE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, WPE))) void enc_wpe(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    const uint64_t t = tile_of_block<2>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 2048, 2048);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 512, 512);
    const uint32_t tid = threadIdx.x;
    u32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * 64 + tid) * 16, 0, kNT));
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * 64 + tid) * 4, 0, kSC0 | kSC1 | kNT);
}
template <int WPE> void add_wpe() {
    char n[96]; snprintf(n, 96, "enc waves_per_eu<=%d (no LDS cap)", WPE); uint64_t t = N / 2048;
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((enc_wpe<WPE>), dim3((unsigned)t), dim3(64), 0, s, d_in, d_packed, t); }, {}, true}); }

static uint64_t checksum(const void* p, uint64_t words, hipStream_t s) {
    static unsigned long long* d_sum = nullptr;
    if (!d_sum) CK(hipMalloc(&d_sum, 8));
    CK(hipMemsetAsync(d_sum, 0, 8, s));
    hipLaunchKernelGGL(checksum_words, dim3(4096), dim3(kBlock), 0, s, static_cast<const uint64_t*>(p), (uint64_t)0, words, d_sum);
    unsigned long long h = 0; CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); return h;
}

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 5, iters = argc > 3 ? atoi(argv[3]) : 2;
    N = 1ull << log2;
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_packed, N / 4)); CK(hipMalloc(&d_out, N));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    hipLaunchKernelGGL((n_to_bits_stream<256, 4, 1, 0, 0, false>), dim3((unsigned)(N / 16384)), dim3(256), 0, s, d_in, d_packed, N / 16384);
    CK(hipStreamSynchronize(s));
    constexpr int A = kSC0 | kSC1 | kNT;
    add<128, 2, 4, 0, 0, A>(13);  // shipped decode
    for (int k : {4, 5}) { add<512, 2, 1, 0, 0, A>(k); add<512, 2, 1, 1, 0, A>(k); }
    for (int k : {2, 3}) { add<1024, 1, 1, 0, 0, A>(k); }
    for (int k : {6, 7, 8}) { add<512, 1, 2, 0, 0, A>(k); }
    add<128, 2, 4, 0, 0, A>(13);
    uint64_t ref_d = 0, ref_e = checksum(d_packed, N / 32, s); bool have = false;
    for (auto& v : vs) {
        if (v.is_enc) {
            CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s));
            v.launch(s); CK(hipGetLastError());
            if (checksum(d_packed, N / 32, s) != ref_e) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
            continue;
        }
        CK(hipMemsetAsync(d_out, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        uint64_t c = checksum(d_out, N / 8, s);
        if (!have) { ref_d = c; have = true; }
        if (c != ref_d) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-44s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 1.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
