// Stress test of the host tier's staging-copy team (cute_nucleotides_amd/csrc/copy_pool.hpp) on the CPU box, built with
// -fsanitize=thread by tests/test_copy_pool.py: random sizes and alignments, warm and fresh-page copies, copies that
// follow each other at once (the helpers are still spinning) and after a pause (they have gone to sleep), pool restarts,
// several calling threads with a pool each.  Every copy is compared byte for byte; guard bytes around the
// destination must survive.  Prints "ok <copies>" and exits 0.
#include "../cute_nucleotides_amd/csrc/copy_pool.hpp"

#include <cstdio>
#include <random>

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
        memset(dst.data() + doff - 64, 0xA5, 64);
        memset(dst.data() + doff + bytes, 0x5A, 64);
        memset(dst.data() + doff, 0, bytes);
        pool.copy(dst.data() + doff, src.data() + so, bytes, fresh);
        if (memcmp(dst.data() + doff, src.data() + so, bytes) != 0) ++bad;
        for (int g = 0; g < 64; ++g)
            if (dst[doff - 64 + g] != 0xA5 || dst[doff + bytes + g] != 0x5A) ++bad;
        if (i % 97 == 0) std::this_thread::sleep_for(std::chrono::microseconds(400));  // longer than the helpers spin
        if (i % 211 == 0) pool.stop();                                                   // restarts at the next copy
        if (i % 389 == 0) pool.set_limit(1 + (int)(rng() % 4));
    }
    return bad;
}

int main(int argc, char** argv) {
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
