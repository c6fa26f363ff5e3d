// latency_lab.hip -- what a SMALL host-tier call costs and how much of it is completion detection.
// The zero-copy path of the host tier (csrc/host_tier.inc host_encode: memcpy into pinned staging, ONE
// generic-kernel launch that reads / writes pinned memory over PCIe, hipStreamSynchronize, memcpy out)
// costs 15-19 us at the reference's bench size (40 000 nt); a resident enqueue + sync costs 13 us.  This
// lab times the same call with different ways of learning that the kernel is done:
//   sync        hipStreamSynchronize (shipped)
//   event       hipEventRecord + hipEventSynchronize
//   query       spin on hipStreamQuery
//   wv32        hipStreamWriteValue32 to a pinned flag + host spin on the flag
//   sigk        a 1-thread signal kernel behind the codec kernel + host spin on the pinned flag
//   fused       the codec kernel itself raises the flag from its last workgroup (threadfence_system +
//               device counter) + host spin
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/latency_lab bench/latency_lab.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"

using namespace cnt;
using clk = std::chrono::steady_clock;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void signal_kernel(volatile uint32_t* flag, uint32_t v) {
    __atomic_store_n(const_cast<uint32_t*>(flag), v, __ATOMIC_RELEASE);  // system scope by default for host memory? made explicit below
    __threadfence_system();
}

// n_to_bits_generic + completion flag raised by the last workgroup to finish
__global__ __launch_bounds__(kBlock) void encode_flag(const uint8_t* __restrict__ n, uint64_t n_len, uint64_t* __restrict__ out, uint64_t n_words,
                                                      unsigned int* counter, uint32_t* flag, uint32_t v) {
    for (uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x; w < n_words; w += (uint64_t)gridDim.x * kBlock) {
        const uint64_t i0 = w << 5;
        uint64_t acc = 0;
        const int m = (n_len - i0) < 32 ? (int)(n_len - i0) : 32;
        for (int k = 0; k < m; k += 4) {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (k + j < m) x |= (uint32_t)n[i0 + k + j] << (8 * j);
            acc |= (uint64_t)__builtin_amdgcn_ubfe(enc4<false>(x), 19, 8) << (2 * k);
        }
        out[w] = acc;
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int prev = atomicAdd(counter, 1u);
        if (prev == gridDim.x - 1) {
            *counter = 0;  // ready for the next launch (stream order)
            __threadfence_system();
            __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void empty_kernel() {}

// one thread per output word, the staging buffer is 16-B aligned and zero-padded to a whole word: two 16-B loads
// issued together (ONE PCIe round trip per thread instead of a chain of byte loads), one 8-B store
__global__ __launch_bounds__(kBlock) void encode_wide(const uint8_t* __restrict__ n, uint64_t* __restrict__ out, uint64_t n_words) {
    const uint64_t w = blockIdx.x * (uint64_t)kBlock + threadIdx.x;
    if (w >= n_words) return;
    const u32x4* p = reinterpret_cast<const u32x4*>(n + 32 * w);
    const u32x4 a = p[0], b = p[1];
    out[w] = (uint64_t)enc16<false>(a) | ((uint64_t)enc16<false>(b) << 32);
}

static inline void spin_until(volatile uint32_t* flag, uint32_t v) {
    while (__atomic_load_n(const_cast<uint32_t*>(flag), __ATOMIC_ACQUIRE) != v) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

int main(int argc, char** argv) {
    const size_t n_len = argc > 1 ? (size_t)atol(argv[1]) : 40000;
    const int iters = argc > 2 ? atoi(argv[2]) : 20000;
    const bool spin_flag = argc > 3 && atoi(argv[3]);
    if (spin_flag) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    const size_t words = (n_len + 31) / 32;
    uint8_t *h_in, *h_out;
    uint32_t* h_flag;
    CK(hipHostMalloc((void**)&h_in, n_len + 64, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_out, words * 8 + 64, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&h_flag, 64, hipHostMallocDefault));
    void *d_in, *d_out, *d_flag;
    CK(hipHostGetDevicePointer(&d_in, h_in, 0)); CK(hipHostGetDevicePointer(&d_out, h_out, 0)); CK(hipHostGetDevicePointer(&d_flag, h_flag, 0));
    unsigned int* d_counter; CK(hipMalloc(&d_counter, 4)); CK(hipMemset(d_counter, 0, 4));
    std::vector<uint8_t> src(n_len), dst(words * 8), want(words * 8);
    for (size_t i = 0; i < n_len; ++i) src[i] = "ATCG"[i & 3];
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const unsigned grid = (unsigned)std::max<size_t>(1, (words + kBlock - 1) / kBlock);
    uint32_t tick = 0;
    *h_flag = 0;
    auto launch_plain = [&] { hipLaunchKernelGGL((n_to_bits_generic<false>), dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)d_in, (uint64_t)n_len, (uint64_t*)d_out, (uint64_t)0, (uint64_t)words); };
    struct Mode { const char* name; int id; };
    const Mode modes[] = {{"sync   hipStreamSynchronize (shipped)", 0}, {"event  record + hipEventSynchronize", 1}, {"query  spin on hipStreamQuery", 2},
                          {"wv32   hipStreamWriteValue32 + host spin", 3}, {"sigk   signal kernel + host spin", 4}, {"fused  last-workgroup flag + host spin", 5},
                          {"empty  empty kernel + hipStreamSynchronize (floor)", 6}, {"wide   16-B-load kernel + hipStreamSynchronize", 7},
                          {"wide+sigk  16-B-load kernel + signal kernel + spin", 8}};
    auto one_call = [&](int mode) {
        memcpy(h_in, src.data(), n_len);
        ++tick;
        switch (mode) {
            case 0: launch_plain(); CK(hipStreamSynchronize(s)); break;
            case 1: launch_plain(); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); break;
            case 2: launch_plain(); while (hipStreamQuery(s) == hipErrorNotReady) {} break;
            case 3: launch_plain(); CK(hipStreamWriteValue32(s, d_flag, tick, 0)); spin_until(h_flag, tick); break;
            case 4: launch_plain(); hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, (volatile uint32_t*)d_flag, tick); spin_until(h_flag, tick); break;
            case 5: hipLaunchKernelGGL(encode_flag, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)d_in, (uint64_t)n_len, (uint64_t*)d_out, (uint64_t)words, d_counter, (uint32_t*)d_flag, tick);
                    spin_until(h_flag, tick); break;
            case 6: hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); CK(hipStreamSynchronize(s)); memcpy(h_out, want.data(), words * 8); break;
            case 7: memset(h_in + n_len, 0, 32); hipLaunchKernelGGL(encode_wide, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)d_in, (uint64_t*)d_out, (uint64_t)words); CK(hipStreamSynchronize(s)); break;
            case 8: memset(h_in + n_len, 0, 32); hipLaunchKernelGGL(encode_wide, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)d_in, (uint64_t*)d_out, (uint64_t)words);
                    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(1), 0, s, (volatile uint32_t*)d_flag, tick); spin_until(h_flag, tick); break;
        }
        memcpy(dst.data(), h_out, words * 8);
    };
    {   // how fast does a kernel read pinned host memory? (the staged kernel alone, HIP events, no host copies)
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        memcpy(h_in, src.data(), n_len); memset(h_in + n_len, 0, 32);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(encode_wide, dim3(grid), dim3(kBlock), 0, s, (const uint8_t*)d_in, (uint64_t*)d_out, (uint64_t)words);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("staged kernel alone over PCIe (pinned in, pinned out): %.2f us per launch = %.1f GB/s of input\n", ms * 100.0, n_len / (ms / 10.0) / 1e6);
        }
        // the same bytes through the DMA engine: hipMemcpyAsync H2D of n_len bytes into device memory
        void* d_tmp; CK(hipMalloc(&d_tmp, n_len + 64));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 10; ++i) CK(hipMemcpyAsync(d_tmp, h_in, n_len, hipMemcpyHostToDevice, s));
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 2) printf("hipMemcpyAsync H2D of the same bytes: %.2f us per copy = %.1f GB/s\n", ms * 100.0, n_len / (ms / 10.0) / 1e6);
        }
        CK(hipFree(d_tmp));
    }
    one_call(0);
    want = dst;
    if (n_len % 4 == 0 && want[0] != 0xD8) { fprintf(stderr, "unexpected encode result\n"); return 2; }
    printf("n_len = %zu nt, %d calls per mode, hipDeviceScheduleSpin = %d\n", n_len, iters, (int)spin_flag);
    for (const Mode& m : modes) {
        for (int i = 0; i < 200; ++i) one_call(m.id);
        memset(dst.data(), 0, dst.size());
        std::vector<double> us;
        us.reserve(iters);
        for (int i = 0; i < iters; ++i) {
            auto t0 = clk::now();
            one_call(m.id);
            us.push_back(std::chrono::duration<double, std::micro>(clk::now() - t0).count());
        }
        if (dst != want) { fprintf(stderr, "MISMATCH in mode %s\n", m.name); return 2; }
        CK(hipStreamSynchronize(s));
        std::sort(us.begin(), us.end());
        double mean = 0; for (double x : us) mean += x; mean /= us.size();
        printf("%-46s mean %7.2f us  median %7.2f  p10 %7.2f  p90 %7.2f  p99 %7.2f\n", m.name, mean, us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], us[us.size() * 99 / 100]);
    }
    return 0;
}
