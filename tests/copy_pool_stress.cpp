// Stress test of the host tier's staging-copy team (hip/copy_pool.hpp) on the CPU box, built with
// -fsanitize=thread by tests/test_copy_pool.py: random sizes and alignments, warm and fresh-page copies, copies that
// follow each other at once (the helpers are still spinning) and after a pause (they have gone to sleep), pool restarts,
// several calling threads with a pool each.  Every copy is compared byte for byte; guard bytes around the
// destination must survive.  Prints "ok <copies>" and exits 0.
// Mode "stall" (ADVICE r03): one helper is parked BETWEEN its loads of a job's fields -- after the pointers and the size,
// before the block size -- while the caller finishes that job (a copy into fresh pages, 2-MiB blocks), frees the buffers
// and runs the next one (a warm copy, 1-MiB blocks).  With round 3's publication order the helper then combined the old
// job's pointers with the new block size, took one block too many of the OLD job (a write into freed memory) and bumped
// the NEW job's completion counter; the test checks that the old destination's canary page stays untouched and that every
// later copy is complete.
#include <atomic>
#include <chrono>
#include <thread>
static std::atomic<int> g_stall_armed{0}, g_stalled{0}, g_release{0};
static void test_stall(int k) {
    int one = 1;
    if (k == 1 && g_stall_armed.compare_exchange_strong(one, 2)) {
        g_stalled.store(1);
        while (!g_release.load()) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}
#define CNT_COPY_POOL_TEST_STALL(k) test_stall(k)
#include "../hip/copy_pool.hpp"

#include <cstdio>
#include <random>

static int run_stall() {
    const size_t big = (size_t)9 << 20;
    std::vector<uint8_t> src(big + 4096, 0x11), src2(big + 4096, 0x22);
    std::vector<uint8_t> old_dst(big + 4096, 0), new_dst(big + 4096, 0);
    CopyPool pool;
    pool.copy(new_dst.data(), src.data(), big, false);  // starts the team; helpers are spinning now
    g_stall_armed.store(1);
    pool.copy(old_dst.data() + 1, src.data(), big, true);  // job G: fresh-page blocks (2 MiB); helper 1 parks between its field loads
    for (int i = 0; i < 20000 && !g_stalled.load(); ++i) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (!g_stalled.load()) { printf("note: the helper never reached the stall point (copy finished without it)\n"); }
    memset(old_dst.data(), 0xEE, old_dst.size());  // the caller owns its buffer again: nothing may write it from here on
    int bad = 0;
    for (int round = 0; round < 3; ++round) {
        memset(new_dst.data(), 0, new_dst.size());
        pool.copy(new_dst.data() + 3, src2.data(), big, false);  // job G+1: warm blocks (1 MiB): twice the block count
        if (round == 0) {
            // the window the advisor describes: the new job's fields are published; let the parked helper go on
            g_release.store(1);
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
        if (memcmp(new_dst.data() + 3, src2.data(), big) != 0) ++bad;
    }
    for (uint8_t b : old_dst) bad += b != 0xEE;
    return bad;
}

static int run_thread(unsigned seed, int copies, size_t max_bytes) {
    std::mt19937_64 rng(seed);
    std::vector<uint8_t> src(max_bytes + 4096), dst(max_bytes + 3 * 4096);
    for (auto& b : src) b = (uint8_t)rng();
    CopyPool pool;
    int bad = 0;
    for (int i = 0; i < copies; ++i) {
        const size_t kind = rng() % 8;
        size_t bytes = kind == 0 ? rng() % 4096 : kind == 1 ? ((size_t)512 << 10) + rng() % 3 - 1 : rng() % max_bytes;
        const size_t so = rng() % 4096, doff = 4096 + rng() % 4096;
        const bool fresh = (rng() & 3) == 0;
        const bool stream = (rng() & 1) == 0;  // non-temporal stores (round 6): any alignment, any length, guards intact
        memset(dst.data() + doff - 64, 0xA5, 64);
        memset(dst.data() + doff + bytes, 0x5A, 64);
        memset(dst.data() + doff, 0, bytes);
        pool.copy(dst.data() + doff, src.data() + so, bytes, fresh, stream);
        if (memcmp(dst.data() + doff, src.data() + so, bytes) != 0) ++bad;
        for (int g = 0; g < 64; ++g)
            if (dst[doff - 64 + g] != 0xA5 || dst[doff + bytes + g] != 0x5A) ++bad;
        if (i % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(400));  // longer than the helpers spin
        if (i % 211 == 0) pool.stop();                                                   // restarts at the next copy
        if (i % 389 == 0) pool.set_limit(1 + (int)(rng() % 4));
        if (i % 173 == 0) {  // round 6: helpers pinned one per cache domain -- here: one per allowed CPU, or not at all; the team restarts
            std::vector<cpu_set_t> doms;
            cpu_set_t allowed;
            CPU_ZERO(&allowed);
            if ((rng() & 1) && sched_getaffinity(0, sizeof allowed, &allowed) == 0)
                for (int c = 0; c < CPU_SETSIZE && doms.size() < 5; ++c)
                    if (CPU_ISSET(c, &allowed)) {
                        cpu_set_t one;
                        CPU_ZERO(&one);
                        CPU_SET(c, &one);
                        doms.push_back(one);
                    }
            pool.set_domains(doms, (size_t)(rng() % 7));
        }
    }
    return bad;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "stall")) {
        int bad = 0;
        for (int i = 0; i < 20; ++i) {
            g_stall_armed.store(0); g_stalled.store(0); g_release.store(0);
            bad += run_stall();
        }
        if (bad) { printf("FAILED: %d bad bytes\n", bad); return 1; }
        printf("ok stall\n");
        return 0;
    }
    {  // an ISOLATED copy of 512 KiB .. 1 MiB never starts the team; the second one of a streak does; a running team is joined
        std::vector<uint8_t> src((size_t)4 << 20, 0x33), dst((size_t)4 << 20, 0);
        {
            CopyPool pool;
            pool.copy(dst.data(), src.data(), (size_t)768 << 10, false);
            if (pool.spawned() != 0) { printf("FAILED: a lone mid-size copy started %d helper threads\n", pool.spawned()); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(5));  // longer than a streak: the next one is isolated again
            pool.copy(dst.data(), src.data(), (size_t)1 << 20, true);
            const int after_two = pool.spawned();
            pool.copy(dst.data() + (2 << 20), src.data(), (size_t)1 << 20, false);  // right behind the previous one: a loop of mid-size calls
            const int after_three = pool.spawned();
            if (after_two != 0 || memcmp(dst.data(), src.data(), (size_t)1 << 20) != 0) { printf("FAILED: two isolated mid-size copies started %d helper threads\n", after_two); return 1; }
            if (after_three < 1 || memcmp(dst.data() + (2 << 20), src.data(), (size_t)1 << 20) != 0) { printf("FAILED: a streak of mid-size copies did not start the team\n"); return 1; }
        }
        CopyPool pool;
        pool.copy(dst.data(), src.data(), (size_t)3 << 20, false);  // a large copy starts it at once
        const int team = pool.spawned();
        pool.copy(dst.data() + 5, src.data(), (size_t)768 << 10, false);  // ... and a mid-size one joins it
        if (team < 1 || pool.spawned() != team || memcmp(dst.data() + 5, src.data(), (size_t)768 << 10) != 0) { printf("FAILED: team %d -> %d\n", team, pool.spawned()); return 1; }
    }
    const int copies = argc > 1 ? atoi(argv[1]) : 600, threads = argc > 2 ? atoi(argv[2]) : 3;
    const size_t max_bytes = (size_t)(argc > 3 ? atoi(argv[3]) : 6) << 20;
    std::vector<int> bad(threads, 0);
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; ++t) ts.emplace_back([&, t] { bad[t] = run_thread(1234u + t, copies, max_bytes); });
    for (auto& t : ts) t.join();
    int total = 0;
    for (int b : bad) total += b;
    if (total) {
        printf("FAILED: %d mismatches\n", total);
        return 1;
    }
    printf("ok %d\n", copies * threads);
    return 0;
}
