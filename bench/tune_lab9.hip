// tune_lab9.hip -- chip-wide read/write PHASE SEPARATION by wall clock (bench only).  Every wave
// reads the constant-rate global counter (s_memrealtime, 100 MHz) and only issues loads while
// (now % PERIOD) is in the read window and stores while it is in the write window, holding K tiles
// of packed output in registers in between -- so HBM sees long pure-read and pure-write bursts
// instead of a fine 4:1 mix, with no inter-workgroup communication at all.  Persistent waves:
// one wave per workgroup, grid = resident waves, wave w takes chunks w, w+G, ...
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab9 bench/tune_lab9.hip
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

__device__ __forceinline__ uint32_t phase_now(uint32_t period) { return (uint32_t)(__builtin_amdgcn_s_memrealtime() % period); }

// K = 1-KiB loads per wave per chunk (chunk = K KiB of ASCII, K*256 B packed).  SYNC 0 = no clock gating (control).
template <int K, int SYNC>
__global__ __launch_bounds__(64) void enc_phase(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_chunks,
                                                uint32_t period, uint32_t read_ticks) {
    constexpr uint32_t CH_IN = K * 1024, CH_OUT = CH_IN / 4;
    const uint32_t lane = threadIdx.x;
    for (uint64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + c * CH_IN, CH_IN);
        const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + c * CH_OUT, CH_OUT);
        if (SYNC) {  // wait for the read window (leave a little room before it closes)
            while (phase_now(period) >= read_ticks - read_ticks / 8) __builtin_amdgcn_s_sleep(8);
        }
        u32x4 v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (k * 64 + lane) * 16, 0, kNT));
        uint32_t o[K];
#pragma unroll
        for (int k = 0; k < K; ++k) o[k] = enc16<false>(v[k]);
        if (SYNC) {
            while (phase_now(period) < read_ticks) __builtin_amdgcn_s_sleep(8);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) __builtin_amdgcn_raw_buffer_store_b32(o[k], rout, (k * 64 + lane) * 4, 0, kSC0 | kSC1 | kNT);
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };
static uint8_t *d_in, *d_packed; static uint64_t N;
static std::vector<Variant> vs;
template <int K, int SYNC> void add(int waves_per_cu, uint32_t period_us10, uint32_t read_frac_pct) {
    char n[128]; snprintf(n, 128, "enc_phase K=%-2d sync=%d waves/CU=%-2d period=%4.1fus read=%u%%", K, SYNC, waves_per_cu, period_us10 / 10.0, read_frac_pct);
    const uint64_t chunks = N / (K * 1024ull); const unsigned g = 256u * waves_per_cu;
    const uint32_t period = period_us10 * 10, read_ticks = period * read_frac_pct / 100;  // 100 MHz ticks: 1 us = 100 ticks
    vs.push_back({n, [chunks, g, period, read_ticks](hipStream_t s) { hipLaunchKernelGGL((enc_phase<K, SYNC>), dim3(g), dim3(64), 0, s, d_in, d_packed, chunks, period, read_ticks); }, {}}); }

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
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_packed, N / 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    CK(hipStreamSynchronize(s));
    vs.push_back({"shipped one-shot B=64 U=2 pairs cap 23", [](hipStream_t st) { const uint64_t t = N / 2048; hipLaunchKernelGGL((n_to_bits_stream<64, 2, 2, kNT, kSC0 | kSC1 | kNT, false>), dim3((unsigned)t), dim3(64), 6912, st, d_in, d_packed, t); }, {}});
    for (int w : {16, 24, 32}) { add<8, 0>(w, 0, 0); add<16, 0>(w, 0, 0); }
    for (int w : {16, 24, 32})
        for (uint32_t p : {100u, 200u, 400u}) { add<8, 1>(w, p, 80); add<16, 1>(w, p, 80); }
    add<16, 1>(24, 200, 70); add<16, 1>(24, 200, 85); add<8, 1>(32, 50, 80); add<16, 1>(32, 800, 80);
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
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-64s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 1.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
