// munmap_lab.cpp -- where the time of "a fresh 1-GiB output per call" goes (bench/host_fresh_lab.py: a decode into a fresh
// numpy array costs 74 ms per call when the array is freed inside the timed loop and 36 ms when it is freed later; into a
// reused buffer 26 ms).  Pure host experiment, with and without the HIP runtime initialised in the process:
//   mmap 1 GiB -> [madvise HUGEPAGE] -> first-touch it with T threads -> munmap, each step timed;
// also AnonHugePages of the region (from /proc/self/smaps) right after the touch, to see whether the advice took.
//   hipcc -O2 -std=c++17 -o bench/munmap_lab bench/munmap_lab.cpp
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

static long anon_huge_kib(void* p) {
    FILE* f = fopen("/proc/self/smaps", "r");
    if (!f) return -1;
    char line[512];
    unsigned long lo = 0, hi = 0;
    bool in = false;
    long kib = -1;
    while (fgets(line, sizeof line, f)) {
        if (sscanf(line, "%lx-%lx ", &lo, &hi) == 2 && strchr(line, '-') && line[0] != 'A') in = lo <= (unsigned long)p && (unsigned long)p < hi;
        if (in && !strncmp(line, "AnonHugePages:", 14)) { kib = atol(line + 14); break; }
    }
    fclose(f);
    return kib;
}

static void touch(uint8_t* p, size_t bytes, int threads) {
    std::vector<std::thread> ts;
    const size_t per = bytes / threads;
    for (int k = 0; k < threads; ++k)
        ts.emplace_back([=] { for (size_t off = per * k; off < per * (k + 1); off += 4096) p[off] = 1; });
    for (auto& t : ts) t.join();
}

static void run(const char* label, size_t bytes, bool huge, int threads) {
    double t_map = 0, t_touch = 0, t_unmap = 0;
    long ahp = 0;
    const int reps = 4;
    for (int r = 0; r < reps; ++r) {
        auto t0 = clk::now();
        uint8_t* p = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (huge) madvise(p, bytes, MADV_HUGEPAGE);
        t_map += ms_since(t0);
        t0 = clk::now();
        touch(p, bytes, threads);
        t_touch += ms_since(t0);
        ahp = anon_huge_kib(p);
        t0 = clk::now();
        munmap(p, bytes);
        t_unmap += ms_since(t0);
    }
    printf("%-44s mmap %6.2f ms  first touch (%2d thr) %7.2f ms  munmap %7.2f ms  AnonHugePages %ld KiB\n", label, t_map / reps, threads, t_touch / reps,
           t_unmap / reps, ahp);
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 30);
    printf("-- before the HIP runtime is initialised\n");
    run("4-KiB pages", bytes, false, 8);
    run("MADV_HUGEPAGE", bytes, true, 8);
    void* d = nullptr;
    if (hipMalloc(&d, 1 << 20) != hipSuccess) { printf("no HIP device\n"); return 0; }
    hipStream_t s;
    (void)hipStreamCreate(&s);
    printf("-- with the HIP runtime initialised (one device allocation, one stream)\n");
    run("4-KiB pages", bytes, false, 8);
    run("MADV_HUGEPAGE", bytes, true, 8);
    run("MADV_HUGEPAGE, 1 touching thread", bytes, true, 1);
    void* h = nullptr;
    (void)hipHostMalloc(&h, 64 << 20, hipHostMallocDefault);
    printf("-- and 64 MiB of pinned host memory allocated\n");
    run("4-KiB pages", bytes, false, 8);
    run("MADV_HUGEPAGE", bytes, true, 8);
    return 0;
}
