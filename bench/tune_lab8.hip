// tune_lab8.hip -- last-percent sweep for encode under the residency cap (bench only): cap x
// cache policy x a few shapes, incl. a "pair-lane" layout (each lane loads 2 adjacent 16-B
// vectors and stores 8 B).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab8 bench/tune_lab8.hip
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

// LAYOUT 0: as shipped (load u at (u*BLOCK+tid)*16, dword store).  LAYOUT 1: pair-lane (U must be 2).
template <int BLOCK, int U, int C, int LAYOUT, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void enc8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    u32x4 v[U];
    if constexpr (LAYOUT == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
        touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * BLOCK + tid) * 4, 0, SAUX);
    } else {
        static_assert(LAYOUT == 0 || U == 2, "pair-lane needs U == 2");
        v[0] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, tid * 32, 0, LAUX));
        v[1] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, tid * 32 + 16, 0, LAUX));
        touch_residency_pad(n_tiles, v[0].x);
        typedef unsigned int vu2 __attribute__((__vector_size__(8)));
        const vu2 o = {enc16<false>(v[0]), enc16<false>(v[1])};
        __builtin_amdgcn_raw_buffer_store_b64(o, rout, tid * 8, 0, SAUX);
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };
static uint8_t *d_in, *d_packed; static uint64_t N;
static std::vector<Variant> vs;
template <int B, int U, int C, int LY, int L, int S> void add(int cap) {
    char n[112]; snprintf(n, 112, "enc B=%-3d U=%d C=%d ly=%d ld=%-2d st=%-2d cap=%-2d", B, U, C, LY, L, S, cap); uint64_t t = N / (B * U * 16);
    size_t lds = cap ? (size_t)(163840 / cap) / 256 * 256 : 0;
    vs.push_back({n, [t, lds](hipStream_t s) { hipLaunchKernelGGL((enc8<B, U, C, LY, L, S>), dim3((unsigned)t), dim3(B), lds, s, d_in, d_packed, t); }, {}}); }

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
    for (int cap : {22, 23, 24}) {
        add<64, 2, 2, 0, 2, 16>(cap); add<64, 2, 2, 0, 2, 19>(cap); add<64, 2, 2, 0, 3, 16>(cap); add<64, 2, 2, 0, 2, 17>(cap); add<64, 2, 2, 0, 18, 16>(cap); add<64, 2, 2, 0, 2, 2>(cap);
        add<64, 2, 2, 1, 2, 16>(cap); add<64, 2, 1, 1, 2, 16>(cap);
    }
    for (int cap : {10, 11, 12}) { add<128, 2, 1, 0, 2, 16>(cap); add<128, 2, 2, 0, 2, 16>(cap); add<128, 2, 1, 1, 2, 16>(cap); add<128, 2, 2, 1, 2, 16>(cap); }
    for (int cap : {5, 6}) { add<256, 2, 1, 0, 2, 16>(cap); add<256, 2, 1, 1, 2, 16>(cap); }
    uint64_t ref = 0; bool have = false;
    for (auto& v : vs) {
        CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        uint64_t c = checksum(d_packed, N / 32, s);
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
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::sort(vs.begin(), vs.end(), [&](const Variant& a, const Variant& b) { return med(a.ms) < med(b.ms); });
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-48s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 1.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
