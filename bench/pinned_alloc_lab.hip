// pinned_alloc_lab.hip -- does the HISTORY of hipHostMalloc / hipHostFree calls decide how fast the DMA engines (and the CPU)
// move data through the pinned buffer that comes out at the end?  Round 6, bench/host_tier_lab.py `pipeline`: the same 16-Mi-nt
// chunk pipeline ran 1-GiB calls in 22.1 ms or in 25.9-31 ms depending only on which SMALLER calls the process had made
// before -- i.e. on how the staging ring had grown (hip/shim_host_ctx.inc DevCtx::ensure frees and re-allocates all slots
// whenever a call needs more).  This lab allocates six 16-MiB pinned buffers (three slots x in / out) after different
// histories and measures, per buffer: H2D and D2H GiB/s of a 16-MiB hipMemcpyAsync (median of 15) and the GiB/s of a
// single-thread memcpy into it.
//
//   hipcc --offload-arch=gfx950 -O2 -o bench/pinned_alloc_lab bench/pinned_alloc_lab.hip && bench/pinned_alloc_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                                \
        }                                                                           \
    } while (0)

static const size_t kBuf = (size_t)16 << 20;
static void* g_dev = nullptr;
static hipStream_t g_s;
static std::vector<char> g_src;

static int node_of(const void* p) {
    void* page = (void*)((uintptr_t)p & ~(uintptr_t)4095);
    int status = -1;
    if (syscall(SYS_move_pages, 0, 1UL, &page, nullptr, &status, 0) != 0) return -1;
    return status;
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

static void measure(const char* label, const std::vector<void*>& bufs, unsigned flags_used) {
    double h2d_min = 1e9, h2d_max = 0, d2h_min = 1e9, d2h_max = 0, cpu_min = 1e9, cpu_max = 0;
    for (void* b : bufs) {
        std::vector<double> up, down, cpu;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        for (int i = 0; i < 17; ++i) {
            float ms = 0;
            CK(hipEventRecord(e0, g_s));
            CK(hipMemcpyAsync(g_dev, b, kBuf, hipMemcpyHostToDevice, g_s));
            CK(hipEventRecord(e1, g_s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) up.push_back(kBuf / (ms * 1e-3) / (double)(1 << 30));
            CK(hipEventRecord(e0, g_s));
            CK(hipMemcpyAsync(b, g_dev, kBuf, hipMemcpyDeviceToHost, g_s));
            CK(hipEventRecord(e1, g_s));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (i >= 2) down.push_back(kBuf / (ms * 1e-3) / (double)(1 << 30));
            const auto t0 = std::chrono::steady_clock::now();
            memcpy(b, g_src.data(), kBuf);
            const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (i >= 2) cpu.push_back(kBuf / s / (double)(1 << 30));
        }
        CK(hipEventDestroy(e0));
        CK(hipEventDestroy(e1));
        const double u = median(up), d = median(down), c = median(cpu);
        h2d_min = std::min(h2d_min, u), h2d_max = std::max(h2d_max, u);
        d2h_min = std::min(d2h_min, d), d2h_max = std::max(d2h_max, d);
        cpu_min = std::min(cpu_min, c), cpu_max = std::max(cpu_max, c);
    }
    printf("{\"history\": \"%s\", \"flags\": %u, \"buffers\": %zu, \"h2d_GiBs\": [%.1f, %.1f], \"d2h_GiBs\": [%.1f, %.1f], \"memcpy_1thread_GiBs\": [%.1f, %.1f], \"node_of_first\": %d, "
           "\"addr_mod_2MiB\": %zu}\n",
           label, flags_used, bufs.size(), h2d_min, h2d_max, d2h_min, d2h_max, cpu_min, cpu_max, node_of(bufs[0]), (size_t)((uintptr_t)bufs[0] & ((2u << 20) - 1)));
    fflush(stdout);
}

static std::vector<void*> alloc_n(size_t bytes, int n, unsigned flags) {
    std::vector<void*> v;
    for (int i = 0; i < n; ++i) {
        void* p = nullptr;
        CK(hipHostMalloc(&p, bytes, flags));
        memset(p, 1, bytes);
        v.push_back(p);
    }
    return v;
}
static void free_all(std::vector<void*>& v) {
    for (void* p : v) CK(hipHostFree(p));
    v.clear();
}

int main() {
    CK(hipSetDevice(0));
    CK(hipStreamCreateWithFlags(&g_s, hipStreamNonBlocking));
    CK(hipMalloc(&g_dev, kBuf));
    g_src.assign(kBuf, 7);
    for (unsigned flags : {(unsigned)hipHostMallocDefault, (unsigned)hipHostMallocNonCoherent}) {
        {  // 1. straight to the final size
            auto v = alloc_n(kBuf, 6, flags);
            measure("fresh: six 16-MiB buffers allocated first thing", v, flags);
            free_all(v);
        }
        {  // 2. the ring's growth under the default chunk rule when calls arrive in increasing size (1 -> 4 -> 16 MiB)
            for (size_t mib : {1, 4}) {
                auto v = alloc_n(mib << 20, 6, flags);
                free_all(v);
            }
            auto v = alloc_n(kBuf, 6, flags);
            measure("grown 1 -> 4 -> 16 MiB, every step freed before the next", v, flags);
            free_all(v);
        }
        {  // 3. slot by slot, as DevCtx::ensure does it: free slot i's old pair, allocate its new pair, next slot
            std::vector<void*> cur = alloc_n((size_t)256 << 10, 6, flags);
            for (size_t kib : {1024, 4096, 16384}) {
                for (size_t i = 0; i < cur.size(); ++i) {
                    CK(hipHostFree(cur[i]));
                    CK(hipHostMalloc(&cur[i], kib << 10, flags));
                    memset(cur[i], 1, kib << 10);
                }
            }
            measure("grown 0.25 -> 1 -> 4 -> 16 MiB slot by slot (free one, allocate one)", cur, flags);
            free_all(cur);
        }
        {  // 4. one slab for the whole ring
            void* slab = nullptr;
            CK(hipHostMalloc(&slab, 6 * kBuf, flags));
            memset(slab, 1, 6 * kBuf);
            std::vector<void*> v;
            for (int i = 0; i < 6; ++i) v.push_back((char*)slab + i * kBuf);
            measure("one 96-MiB slab cut into six", v, flags);
            CK(hipHostFree(slab));
        }
        {  // 5. malloc'ed memory registered in place (2-MiB aligned, huge-page advised)
            void* raw = nullptr;
            if (posix_memalign(&raw, 2u << 20, 6 * kBuf) == 0) {
                madvise(raw, 6 * kBuf, 14 /* MADV_HUGEPAGE */);
                memset(raw, 1, 6 * kBuf);
                if (hipHostRegister(raw, 6 * kBuf, hipHostRegisterDefault) == hipSuccess) {
                    std::vector<void*> v;
                    for (int i = 0; i < 6; ++i) v.push_back((char*)raw + i * kBuf);
                    measure("posix_memalign(2 MiB) + MADV_HUGEPAGE + hipHostRegister, cut into six", v, flags);
                    CK(hipHostUnregister(raw));
                }
                free(raw);
            }
        }
    }
    return 0;
}
