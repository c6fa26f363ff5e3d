// tune_lab12.hip -- what the encode arithmetic costs.  hipcc turns enc_gather's
// (y<<6|y), (u<<12|u) into ONE v_mul_lo_u32 by 0x41041 (disjoint bits: OR == ADD == multiply; the same
// identity as the reference's n_to_bits_mul, n_to_bits.rs:223-231) -- a quarter-rate instruction, 8 per lane
// and tile.  A/B against the two full-rate v_lshl_or_b32 the source spells (forced with inline asm) and
// against a v_perm_b32-only gather, all in the shipped shape (64 thr x 2 loads, XCD pairs, 23 wg/CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab12 bench/tune_lab12.hip
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

__device__ __forceinline__ uint32_t gather_shift_or(uint32_t y) {  // packed byte at bits 19..26
    uint32_t u, r;
    asm("v_lshl_or_b32 %0, %1, 6, %1" : "=v"(u) : "v"(y));
    asm("v_lshl_or_b32 %0, %1, 12, %1" : "=v"(r) : "v"(u));
    return r;
}

// MODE 0: shipped source (compiler's v_mul_lo_u32); 1: forced v_lshl_or_b32 pairs; 2: no arithmetic (xor of the
// four dwords -- the access pattern alone)
template <int MODE>
__device__ __forceinline__ uint32_t pack16(u32x4 q) {
    if constexpr (MODE == 0) return enc16<false>(q);
    if constexpr (MODE == 2) return q.x ^ q.y ^ q.z ^ q.w;
    uint32_t b0 = __builtin_amdgcn_ubfe(gather_shift_or(q.x & 0x06060606u), 19, 8);
    uint32_t b1 = __builtin_amdgcn_ubfe(gather_shift_or(q.y & 0x06060606u), 19, 8);
    uint32_t b2 = __builtin_amdgcn_ubfe(gather_shift_or(q.z & 0x06060606u), 19, 8);
    uint32_t b3 = gather_shift_or(q.w & 0x06060606u) << 5;
    return b0 | (b1 << 8) | (b2 << 16) | (b3 & 0xFF000000u);
}

template <int MODE>
__global__ __launch_bounds__(64) void enc12(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    extern __shared__ uint32_t pad[];
    const uint64_t t = tile_of_block<2>(blockIdx.x, (uint32_t)n_tiles, 3 /* log2 of the 8 XCDs of an MI355X in SPX mode */);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 2048, 2048);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 512, 512);
    const uint32_t tid = threadIdx.x;
    u32x4 v[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * 64 + tid) * 16, 0, kNT));
    if (n_tiles == 0xFFFFFFFFFFFFFFFFull) pad[tid] = v[0].x;
#pragma unroll
    for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b32(pack16<MODE>(v[u]), rout, (u * 64 + tid) * 4, 0, kSC0 | kSC1 | kNT);
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; bool check; };
static uint8_t *d_in, *d_packed;
static uint64_t N;
static std::vector<Variant> vs;
template <int MODE> void add(const char* what, int cap) {
    char n[96]; snprintf(n, 96, "enc %-28s cap=%d", what, cap); uint64_t t = N / 2048;
    size_t lds = (size_t)(163840 / cap) / 256 * 256;
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((enc12<MODE>), dim3((unsigned)t), dim3(64), lds, s, d_in, d_packed, t); }, {}, MODE != 2}); }

static uint64_t checksum(const void* p, uint64_t words, hipStream_t s) {
    static unsigned long long* d_sum = nullptr;
    if (!d_sum) CK(hipMalloc(&d_sum, 8));
    CK(hipMemsetAsync(d_sum, 0, 8, s));
    hipLaunchKernelGGL(checksum_words, dim3(4096), dim3(kBlock), 0, s, static_cast<const uint64_t*>(p), (uint64_t)0, words, d_sum);
    unsigned long long h = 0; CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); return h;
}

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 7, iters = argc > 3 ? atoi(argv[3]) : 2;
    N = 1ull << log2;
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_packed, N / 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    CK(hipStreamSynchronize(s));
    for (int rep = 0; rep < 2; ++rep)
        for (int cap : {23, 22}) { add<0>("v_mul_lo_u32 (shipped)", cap); add<1>("v_lshl_or_b32 x2 (forced)", cap); add<2>("no arithmetic (xor)", cap); }
    uint64_t ref = 0; bool have = false;
    for (auto& v : vs) {
        CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        if (!v.check) continue;
        const uint64_t c = checksum(d_packed, N / 32, s);
        if (!have) { ref = c; have = true; }
        if (c != ref) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
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
