// fresh_pages_lab.cpp -- the host tier's copy-out into a FRESH output buffer (the reference's rule: every call
// returns a new Vec) is page-fault-bound: 13.6 GiB/s for a 1-GiB decode against 42 GiB/s into a reused buffer
// (profiles/r02_bench_final.json host_tier).  Pure host experiment: how fast can `threads` threads copy 1 GiB
// from a warm source into freshly mmap'ed anonymous memory, and what do MADV_HUGEPAGE / MADV_POPULATE_WRITE /
// more threads buy?     g++ -O2 -std=c++17 -pthread -o bench/fresh_pages_lab bench/fresh_pages_lab.cpp
#include <sys/mman.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
using clk = std::chrono::steady_clock;

static void par(int threads, size_t bytes, const std::function<void(size_t, size_t)>& fn) {
    std::vector<std::thread> ts;
    const size_t per = ((bytes + threads - 1) / threads + (2u << 20) - 1) / (2u << 20) * (2u << 20);
    for (int k = 0; k < threads; ++k) {
        const size_t lo = std::min(bytes, per * k), hi = std::min(bytes, per * (k + 1));
        if (hi > lo) ts.emplace_back([=, &fn] { fn(lo, hi); });
    }
    for (auto& t : ts) t.join();
}

int main(int argc, char** argv) {
    const size_t bytes = (size_t)1 << (argc > 1 ? atoi(argv[1]) : 30);
    uint8_t* src = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    par(16, bytes, [&](size_t lo, size_t hi) { memset(src + lo, 0x41, hi - lo); });
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    char thp[128] = "?";
    if (f) { if (!fgets(thp, sizeof thp, f)) thp[0] = 0; fclose(f); }
    printf("bytes = %zu, transparent_hugepage/enabled: %s", bytes, thp);
    struct Case { const char* name; int advise; int populate_threads; int copy_threads; };
    const Case cases[] = {{"plain, 4 copy threads (shipped)", 0, 0, 4}, {"plain, 8 copy threads", 0, 0, 8}, {"plain, 16 copy threads", 0, 0, 16},
                          {"MADV_HUGEPAGE, 4 copy threads", 1, 0, 4}, {"MADV_HUGEPAGE, 8 copy threads", 1, 0, 8},
                          {"POPULATE_WRITE x16, 4 copy threads", 0, 16, 4}, {"POPULATE_WRITE x8, 4 copy threads", 0, 8, 4},
                          {"HUGEPAGE + POPULATE_WRITE x8, 4 copy threads", 1, 8, 4}, {"reused (warm) destination, 4 copy threads", -1, 0, 4}};
    uint8_t* warm = (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    par(16, bytes, [&](size_t lo, size_t hi) { memset(warm + lo, 1, hi - lo); });
    for (const Case& c : cases) {
        double best = 1e30, sum = 0;
        const int reps = 3;
        for (int r = 0; r < reps; ++r) {
            uint8_t* dst = c.advise < 0 ? warm : (uint8_t*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            auto t0 = clk::now();
            if (c.advise > 0) madvise(dst, bytes, MADV_HUGEPAGE);
            if (c.populate_threads) par(c.populate_threads, bytes, [&](size_t lo, size_t hi) { if (madvise(dst + lo, hi - lo, MADV_POPULATE_WRITE)) perror("populate"); });
            par(c.copy_threads, bytes, [&](size_t lo, size_t hi) { memcpy(dst + lo, src + lo, hi - lo); });
            const double s = std::chrono::duration<double>(clk::now() - t0).count();
            best = std::min(best, s); sum += s;
            if (dst[bytes - 1] != 0x41) return 2;
            if (c.advise >= 0) munmap(dst, bytes);
        }
        printf("%-48s mean %8.2f ms = %6.2f GiB/s   best %6.2f GiB/s\n", c.name, sum / reps * 1e3, bytes / (sum / reps) / (1 << 30), bytes / best / (1 << 30));
    }
    return 0;
}
