// copy_placement_lab.cpp -- WHERE the host tier's copy threads sit decides how fast they copy (round 6).
// bench/host_tier_lab.py `history` found the 1-GiB host-slice calls in exactly two states, 22.1 ms and 25.9 ms, chosen per
// process and unrelated to the staging ring's DMA speed (bench/pinned_alloc_lab.hip: 50.5 GiB/s after every allocation history).
// A calling thread + 3 helpers copy 20 MiB per 16-Mi-nt chunk; on an EPYC every CCD (8 cores sharing an L3) reaches memory
// through ONE link of its own, so four threads inside one CCD share what four threads on four CCDs each have for themselves.
// This lab pins T copy threads by four rules and measures the GB/s of the host tier's own copy pattern (16-MiB pieces of a
// 1-GiB source into a 48-MiB ring, 1-MiB blocks dealt round-robin):
//     same-ccd      T distinct cores of ONE L3 domain
//     smt-pairs     T/2 cores of one L3 domain, both hardware threads of each
//     one-per-ccd   one core in each of T different L3 domains (of one NUMA node)
//     unpinned      wherever the scheduler puts them (what the library did until round 6), five trials: the spread IS the finding
//
//   g++ -O2 -pthread -o bench/copy_placement_lab bench/copy_placement_lab.cpp && bench/copy_placement_lab [node]
#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <vector>

static std::vector<int> parse_list(const std::string& s) {
    std::vector<int> out;
    const char* p = s.c_str();
    while (*p) {
        char* end = nullptr;
        long lo = strtol(p, &end, 10);
        if (end == p) break;
        long hi = lo;
        p = end;
        if (*p == '-') {
            hi = strtol(p + 1, &end, 10);
            p = end;
        }
        for (long c = lo; c <= hi; ++c) out.push_back((int)c);
        while (*p == ',' || *p == '\n' || *p == ' ') ++p;
    }
    return out;
}
static std::string read_file(const std::string& path) {
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return "";
    char buf[8192];
    size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    return buf;
}

static double run(const std::vector<int>& cpus /* empty: unpinned */, int T, const uint8_t* src, uint8_t* ring, size_t total, size_t chunk) {
    const size_t blk = 1 << 20;
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    std::vector<std::thread> ts;
    for (int t = 0; t < T; ++t)
        ts.emplace_back([&, t] {
            if (!cpus.empty()) {
                cpu_set_t set;
                CPU_ZERO(&set);
                CPU_SET(cpus[t % cpus.size()], &set);
                pthread_setaffinity_np(pthread_self(), sizeof set, &set);
            }
            ready.fetch_add(1);
            while (!go.load()) {
            }
            for (size_t c = 0; c < total / chunk; ++c) {
                uint8_t* d = ring + (c % 3) * chunk;
                for (size_t b = t; b < chunk / blk; b += T) memcpy(d + b * blk, src + c * chunk + b * blk, blk);
            }
        });
    while (ready.load() < T) {
    }
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true);
    for (auto& t : ts) t.join();
    return total / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 1e9;
}

int main(int argc, char** argv) {
    const int node = argc > 1 ? atoi(argv[1]) : 0;
    const std::vector<int> node_cpus = parse_list(read_file("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist"));
    if (node_cpus.empty()) {
        fprintf(stderr, "no such NUMA node\n");
        return 1;
    }
    // L3 domains -> cores -> hardware threads
    std::map<std::string, std::map<std::string, std::vector<int>>> l3;
    for (int c : node_cpus) {
        const std::string base = "/sys/devices/system/cpu/cpu" + std::to_string(c);
        l3[read_file(base + "/cache/index3/shared_cpu_list")][read_file(base + "/topology/thread_siblings_list")].push_back(c);
    }
    printf("{\"node\": %d, \"cpus\": %zu, \"l3_domains\": %zu, \"cores_in_first_domain\": %zu}\n", node, node_cpus.size(), l3.size(), l3.begin()->second.size());
    // run from a thread bound to the node, so that source and ring are first touched there
    cpu_set_t nodeset;
    CPU_ZERO(&nodeset);
    for (int c : node_cpus) CPU_SET(c, &nodeset);
    sched_setaffinity(0, sizeof nodeset, &nodeset);
    const size_t total = (size_t)1 << 30, chunk = (size_t)16 << 20;
    uint8_t* src = (uint8_t*)aligned_alloc(2 << 20, total);
    uint8_t* ring = (uint8_t*)aligned_alloc(2 << 20, 3 * chunk);
    memset(src, 1, total);
    memset(ring, 2, 3 * chunk);
    for (int T : {2, 4, 8}) {
        std::vector<int> same, pairs, spread;
        for (auto& core : l3.begin()->second) {
            if ((int)same.size() < T) same.push_back(core.second[0]);
            if ((int)pairs.size() < T)
                for (int c : core.second)
                    if ((int)pairs.size() < T) pairs.push_back(c);
        }
        for (auto& dom : l3)
            if ((int)spread.size() < T) spread.push_back(dom.second.begin()->second[0]);
        struct Rule {
            const char* name;
            std::vector<int> cpus;
        } rules[] = {{"same-ccd", same}, {"smt-pairs", pairs}, {"one-per-ccd", spread}};
        for (auto& r : rules) {
            if ((int)r.cpus.size() < T) continue;
            run(r.cpus, T, src, ring, total, chunk);
            std::vector<double> v;
            for (int i = 0; i < 3; ++i) v.push_back(run(r.cpus, T, src, ring, total, chunk));
            std::sort(v.begin(), v.end());
            printf("{\"threads\": %d, \"rule\": \"%s\", \"GBs_median\": %.1f, \"GBs_min\": %.1f, \"GBs_max\": %.1f}\n", T, r.name, v[1], v[0], v[2]);
            fflush(stdout);
        }
        std::vector<double> v;
        for (int i = 0; i < 5; ++i) v.push_back(run({}, T, src, ring, total, chunk));
        printf("{\"threads\": %d, \"rule\": \"unpinned\", \"GBs_trials\": [%.1f, %.1f, %.1f, %.1f, %.1f]}\n", T, v[0], v[1], v[2], v[3], v[4]);
        fflush(stdout);
    }
    return 0;
}
