// tune_lab4.hip -- persistent workgroups with explicit register prefetch (bench only): each
// workgroup walks tiles b, b+G, b+2G, ... and issues the loads of its next tile before it
// converts and stores the current one, so every resident wave always has a load in flight.
// Compared against the shipped one-tile-per-workgroup kernels in the same process.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab4 bench/tune_lab4.hip
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

template <int BLOCK, int U, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void enc_persist(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint32_t tid = threadIdx.x;
    uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    u32x4 v[U];
    {
        const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
    }
    for (;;) {
        const uint64_t tn = t + gridDim.x;
        const bool more = tn < n_tiles;
        u32x4 w[U];
        if (more) {
            const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + tn * TILE_IN, TILE_IN);
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
        }
        const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * BLOCK + tid) * 4, 0, SAUX);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[u];
        t = tn;
    }
}

template <int BLOCK, int U, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void dec_persist(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint32_t tid = threadIdx.x;
    uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    uint32_t x[U];
    {
        const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN);
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4, 0, LAUX);
    }
    for (;;) {
        const uint64_t tn = t + gridDim.x;
        const bool more = tn < n_tiles;
        uint32_t y[U];
        if (more) {
            const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + tn * TILE_IN, TILE_IN);
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4, 0, LAUX);
        }
        const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x[u])), rout, (u * BLOCK + tid) * 16, 0, SAUX);
        if (!more) break;
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = y[u];
        t = tn;
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; bool is_enc; };
static uint8_t *d_in, *d_packed, *d_out;
static uint64_t N;
static std::vector<Variant> vs;

template <int B, int U, int L, int S> void add_enc_p(int wg_per_cu) { char n[96]; snprintf(n, 96, "enc persist B=%-4d U=%d ld=%-2d st=%-2d wg/CU=%d", B, U, L, S, wg_per_cu);
    uint64_t t = N / (B * U * 16); unsigned g = 256u * wg_per_cu;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((enc_persist<B, U, L, S>), dim3(g), dim3(B), 0, s, d_in, d_packed, t); }, {}, true}); }
template <int B, int U, int L, int S> void add_dec_p(int wg_per_cu) { char n[96]; snprintf(n, 96, "dec persist B=%-4d U=%d ld=%-2d st=%-2d wg/CU=%d", B, U, L, S, wg_per_cu);
    uint64_t t = N / (B * U * 16); unsigned g = 256u * wg_per_cu;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((dec_persist<B, U, L, S>), dim3(g), dim3(B), 0, s, d_packed, d_out, t); }, {}, false}); }
template <int B, int U, int C, int L, int S> void add_enc_1() { char n[96]; snprintf(n, 96, "enc oneshot B=%-4d U=%d C=%d ld=%-2d st=%-2d", B, U, C, L, S); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((n_to_bits_stream<B, U, C, L, S, false>), dim3((unsigned)t), dim3(B), 0, s, d_in, d_packed, t); }, {}, true}); }
template <int B, int U, int C, int L, int S> void add_dec_1() { char n[96]; snprintf(n, 96, "dec oneshot B=%-4d U=%d C=%d ld=%-2d st=%-2d", B, U, C, L, S); uint64_t t = N / (B * U * 16);
    vs.push_back({n, [t](hipStream_t s) { hipLaunchKernelGGL((bits_to_n_stream<B, U, C, L, S>), dim3((unsigned)t), dim3(B), 0, s, d_packed, d_out, t); }, {}, false}); }

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
    CK(hipStreamSynchronize(s));
    add_enc_1<64, 2, 2, 2, 16>(); add_enc_1<256, 1, 1, 2, 16>();
    for (int k : {4, 8, 16, 32}) { add_enc_p<64, 1, 2, 16>(k); add_enc_p<64, 2, 2, 16>(k); }
    for (int k : {2, 4, 8}) { add_enc_p<256, 1, 2, 16>(k); add_enc_p<256, 2, 2, 16>(k); add_enc_p<512, 1, 2, 16>(k / 2 ? k / 2 : 1); add_enc_p<1024, 1, 2, 16>(k / 4 ? k / 4 : 1); }
    add_dec_1<128, 2, 1, 0, 19>();
    for (int k : {4, 8, 16, 32}) { add_dec_p<64, 1, 0, 19>(k); add_dec_p<64, 2, 0, 19>(k); }
    for (int k : {2, 4, 8, 16}) { add_dec_p<128, 1, 0, 19>(k); add_dec_p<128, 2, 0, 19>(k); add_dec_p<256, 1, 0, 19>(k / 2 ? k / 2 : 1); add_dec_p<256, 2, 0, 19>(k / 2 ? k / 2 : 1); }
    uint64_t ref_enc = 0, ref_dec = 0; bool he = false, hd = false;
    hipLaunchKernelGGL((n_to_bits_stream<256, 4, 1, 0, 0, false>), dim3((unsigned)(N / 16384)), dim3(256), 0, s, d_in, d_packed, N / 16384);
    for (auto& v : vs) {
        if (v.is_enc) CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s)); else CK(hipMemsetAsync(d_out, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        uint64_t c = v.is_enc ? checksum(d_packed, N / 32, s) : checksum(d_out, N / 8, s);
        uint64_t& ref = v.is_enc ? ref_enc : ref_dec; bool& have = v.is_enc ? he : hd;
        if (!have) { ref = c; have = true; }
        if (c != ref) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
    }
    if (checksum(d_in, N / 8, s) != ref_dec) { fprintf(stderr, "decode != input\n"); return 3; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms / iters);
        }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::sort(vs.begin(), vs.end(), [&](const Variant& a, const Variant& b) { if (a.is_enc != b.is_enc) return a.is_enc; return med(a.ms) < med(b.ms); });
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-52s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 1.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
