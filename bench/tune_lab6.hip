// tune_lab6.hip -- decode with run-ahead prefetch workgroups (bench only).  Super-groups of 136
// workgroups: 128 consumers (tile = 4 KiB of ASCII = 1 KiB packed; 16 per XCD, block b on XCD
// b%8) + 8 prefetchers (one per XCD) that pull into their XCD's L2 the packed KiB of the 16
// tiles the same XCD will decode A super-groups later.  Idea: consumers then see L2-hit
// latency, live shorter, and can use one store per lane (the write-only optimum).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab6 bench/tune_lab6.hip
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

// CB x U x 16 = 4096 nt per consumer tile.  PF: 0 = no prefetchers in the grid, 1 = prefetchers present.
template <int CB, int U, int A, int LAUX, int SAUX, int PLAUX>
__global__ __launch_bounds__(256) void dec_pf(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles, uint8_t* __restrict__ sink) {
    static_assert(CB * U * 16 == 4096, "consumer tile must be 4 KiB of ASCII");
    extern __shared__ uint32_t pad[];
    const uint64_t b = blockIdx.x;
    const uint64_t sg = b / 136, r = b % 136;
    const uint32_t tid = threadIdx.x;
    if (r >= 128) {  // prefetcher for XCD x = r - 128 (block b is on XCD b%8 = x because 136 % 8 == 0)
        const uint64_t x = r - 128;
        const uint64_t tsg = sg + A;
        if ((tsg + 1) * 128 > n_tiles) return;
        u32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // wave w, load k -> tile index j = x + 8*(4*k + w) of the target super-group
            const uint32_t w = tid >> 6, lane = tid & 63;
            const uint64_t tile = tsg * 128 + x + 8 * (4 * k + w);
            const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + tile * 1024, 1024);
            acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane * 16, 0, PLAUX));
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345679u) reinterpret_cast<u32x4*>(sink)[tid] = acc;  // keep the loads
        return;
    }
    if (tid >= CB) return;
    const uint64_t t = sg * 128 + r;  // consumer tile; runs on XCD r%8
    if (t >= n_tiles) return;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 1024, 1024);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 4096, 4096);
    uint32_t xw[U];
#pragma unroll
    for (int u = 0; u < U; ++u) xw[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * CB + tid) * 4, 0, LAUX);
    if (n_tiles == ~0ull) pad[tid] = xw[0];
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(xw[u])), rout, (u * CB + tid) * 16, 0, SAUX);
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; };
static uint8_t *d_in, *d_packed, *d_out, *d_sink;
static uint64_t N;
static std::vector<Variant> vs;

template <int CB, int U, int A, int L, int S, int PL> void add(int cap) {
    char n[128]; snprintf(n, 128, "dec+pf CB=%-3d U=%d A=%-2d ld=%-2d st=%-2d pld=%-2d cap=%d", CB, U, A, L, S, PL, cap);
    const uint64_t tiles = N / 4096, groups = (tiles + 127) / 128;
    const unsigned grid = (unsigned)(groups * 136);
    const size_t lds = cap ? (size_t)(163840 / cap) / 256 * 256 : 0;
    vs.push_back({n, [tiles, grid, lds](hipStream_t s) { hipLaunchKernelGGL((dec_pf<CB, U, A, L, S, PL>), dim3(grid), dim3(CB), lds, s, d_packed, d_out, tiles, d_sink); }, {}}); }
// note: launched with CB threads, so prefetcher blocks have CB threads too: with CB=256 they cover 16 tiles, with CB=128 only waves 0,1

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
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_packed, N / 4)); CK(hipMalloc(&d_out, N)); CK(hipMalloc(&d_sink, 1 << 20));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    hipLaunchKernelGGL((n_to_bits_stream<256, 4, 1, 0, 0, false>), dim3((unsigned)(N / 16384)), dim3(256), 0, s, d_in, d_packed, N / 16384);
    CK(hipStreamSynchronize(s));
    const uint64_t ref = checksum(d_in, N / 8, s);
    // reference points: the shipped decode kernel, capped and not
    vs.push_back({"shipped B=128 U=2 pairs cap 13", [](hipStream_t st) { const uint64_t t = N / 4096; hipLaunchKernelGGL((bits_to_n_stream<128, 2, 2, 0, 19>), dim3((unsigned)t), dim3(128), 12544, st, d_packed, d_out, t); }, {}});
    vs.push_back({"shipped B=128 U=2 pairs no cap", [](hipStream_t st) { const uint64_t t = N / 4096; hipLaunchKernelGGL((bits_to_n_stream<128, 2, 2, 0, 19>), dim3((unsigned)t), dim3(128), 0, st, d_packed, d_out, t); }, {}});
    for (int cap : {0, 7, 6}) {
        add<256, 1, 4, 0, 19, 0>(cap); add<256, 1, 8, 0, 19, 0>(cap); add<256, 1, 16, 0, 19, 0>(cap); add<256, 1, 32, 0, 19, 0>(cap); add<256, 1, 64, 0, 19, 0>(cap);
    }
    add<256, 1, 16, 0, 19, 16>(0); add<256, 1, 16, 0, 19, 1>(0); add<256, 1, 16, 16, 19, 0>(0); add<256, 1, 16, 0, 18, 0>(0);
    for (auto& v : vs) {
        CK(hipMemsetAsync(d_out, 0xFF, 1 << 20, s));
        v.launch(s); CK(hipGetLastError());
        if (checksum(d_out, N / 8, s) != ref) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
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
        printf("%-56s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], 1.25 * N / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
