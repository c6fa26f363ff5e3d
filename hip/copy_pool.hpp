// copy_pool.hpp -- the host tier's staging-copy team.  Plain C++17, no HIP: included by cute_nt.hip (inside its anonymous
// namespace, through shim_host_ctx.inc) and, on its own, by tests/copy_pool_stress.cpp, which hammers it under
// ThreadSanitizer on the CPU box.
#pragma once
// (cute_nt.hip includes every header below at file scope BEFORE it opens its namespace: the lines are no-ops there)
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#ifndef CNT_COPY_POOL_TEST_STALL
#define CNT_COPY_POOL_TEST_STALL(k) ((void)0)
#endif

// ---- per-thread host-copy helpers ---------------------------------------------------
// The host tier stages caller memory through pinned buffers (measured on MI355X / PCIe Gen5,
// profiles/r01_host_tier_lab.log: pageable hipMemcpyAsync runs at 43 GB/s only after the runtime
// has pinned the caller's pages and at 8-14 GB/s on first touch; explicit staging is 20-25 GB/s
// with one copying thread and ~40 GB/s with four, cold or warm).  A small team of helper threads per calling
// thread does the staging copies with it; CNT_HOST_COPY_THREADS (default 6, 1 = no helper threads at all) sizes the team
// of warm copies, the caller included; copies into fresh pages use twice that team, so a calling thread owns up to
// 2 x team - 1 helper threads (11 at the default), each spinning for up to 150 us after a copy before it sleeps.
// memcpy with NON-TEMPORAL stores: the destination is not read by a CPU again soon -- the pinned staging ring, which the DMA
// engine reads next, or an output larger than any cache -- so its lines need neither be fetched for ownership first (a third
// of a plain copy's memory traffic) nor linger dirty in some core's L3, where the device's reads have to find them.
// Lab-selectable (CNT_HOST_NT) and OFF: on the GPU box's host it loses where it matters (1 GiB: 24.1-24.6 ms against 22.3-22.8
// with plain stores; profiles/r06_host_tier.md 4) -- the ring's lines are read by the DMA engine out of the cache hierarchy fast enough.
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void copy_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
    if (head > n) head = n;
    memcpy(dst, src, head);
    dst += head, src += head, n -= head;
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
    }
    _mm_sfence();
    memcpy(dst + i, src + i, n - i);
}
inline bool cpu_has_avx2() {
    static const bool v = __builtin_cpu_supports("avx2");
    return v;
}
#endif
inline void copy_block(uint8_t* dst, const uint8_t* src, size_t n, bool stream_stores) {
#if defined(__x86_64__)
    if (stream_stores && n >= 4096 && cpu_has_avx2()) return copy_stream_avx2(dst, src, n);
#endif
    (void)stream_stores;
    memcpy(dst, src, n);
}

class CopyPool {
   public:
    ~CopyPool() { stop(); }
    // Copies INTO FRESH PAGES (the copy-out of a call whose output has never been touched) use the whole pool, every
    // other copy half of it: each first touch is a page fault, faults parallelise, and 8 threads take a 1-GiB decode into
    // a fresh buffer from 35 to 27 ms (profiles/r03_host_pipeline_slots.jsonl).
    //
    // The team is built for copies that take 50-400 us each, dozens of times per call: a copy is cut into blocks
    // that the caller and the helpers STEAL from one counter (a helper that wakes up late finds nothing left and costs
    // nothing), and a helper that has run dry spins on the job generation for kSpinUs before it goes to sleep on the
    // condition variable -- inside a pipelined call it never sleeps, so a copy starts within a microsecond instead of the
    // 20-50 us of a condition-variable wake-up (which made 8 threads SLOWER than 4 in the first version of this pool).
    void copy(uint8_t* dst, const uint8_t* src, size_t bytes, bool fresh_pages = false, bool stream_stores = false) {
        // small copies never wake a sleeping team, and an ISOLATED one never starts it: below kMinPar always the caller's
        // memcpy; up to kSmallCopy (the band the zero-copy small calls fall into) a team that is already running is joined,
        // and one that is not is started only by the SECOND such copy within kStreakUs -- a loop of mid-size calls, where the
        // spinning helpers take 16 % off a 2^20-nt decode (profiles/r04_host_small_copies.md) -- so that a lone mid-size
        // call does not pay for creating up to seven threads that then spin for 150 us each
        if (bytes < kMinPar) {
            copy_block(dst, src, bytes, stream_stores);
            return;
        }
        if (bytes <= kSmallCopy && !started_) {
            if (std::chrono::steady_clock::now() - last_mid_copy_ >= std::chrono::microseconds(kStreakUs)) {
                copy_block(dst, src, bytes, stream_stores);
                last_mid_copy_ = std::chrono::steady_clock::now();  // the END of this copy: the streak is the gap between calls
                return;
            }
        }
        const int all = threads();
        const int T = fresh_pages ? all : std::max(1, (all + 1) / 2);  // warm copies use the configured team, fresh ones twice that
        if (T <= 1) {
            copy_block(dst, src, bytes, stream_stores);
            return;
        }
        const uint64_t g = gen_.load(std::memory_order_relaxed) + 1;
        // blocks are cut on multiples of the block size of the DESTINATION address (the first one is short): with 2-MiB
        // blocks a huge page of a fresh output is faulted in by ONE thread instead of being fought over by eight
        // (round 4) copies of up to 1 MiB -- the chunks of mid-size calls, 2^20..2^22 nt -- in 256-KiB blocks: with 1-MiB blocks a
        // 1-MiB copy was ONE block, i.e. one thread (larger copies keep 1-MiB blocks: 256-KiB blocks cost 4-MiB copies 5-12 %)
        const size_t blk = fresh_pages ? kFreshBlock : bytes <= small_block_max() ? kSmallBlock : warm_block();
        const size_t skew = reinterpret_cast<uintptr_t>(dst) & (blk - 1);
        const uint64_t nblocks = (skew + bytes + blk - 1) / blk;
        // Publication order: the block counter moves to generation g FIRST -- from here on nobody can take a
        // block of the previous job -- and only then are the job's fields and its completion counter rewritten.  A helper
        // that is still holding generation g-1 and reads any field written below is ordered after the counter's bump (the
        // fence here pairs with the acquire fence in run()), finds generation g in the counter and returns without touching
        // anything; before the bump every field it can see belongs to g-1, whose blocks are all taken.  (Round 3 wrote the
        // block size and reset done_ BEFORE the bump: a helper preempted between its field loads could then combine job
        // g-1's pointers with job g's smaller block size, compute a larger block count, win one more block of the old job
        // and add to the new job's completion counter.)
        next_.store(g << 32, std::memory_order_seq_cst);
        std::atomic_thread_fence(std::memory_order_seq_cst);
        done_.store(0, std::memory_order_relaxed);
        job_dst_.store(dst, std::memory_order_relaxed);
        job_src_.store(src, std::memory_order_relaxed);
        job_bytes_.store(bytes, std::memory_order_relaxed);
        job_blk_.store(blk, std::memory_order_relaxed);
        job_nblocks_.store(nblocks, std::memory_order_relaxed);
        job_team_.store(T, std::memory_order_relaxed);
        job_stream_.store(stream_stores, std::memory_order_relaxed);
        gen_.store(g, std::memory_order_seq_cst);
        // helpers that are still spinning (the previous copy was < 150 us ago) join at once; sleeping ones are woken only for
        // copies worth a 20-50 us wake-up -- a 1-MiB copy of an isolated small call is the caller's alone, as it always was
        if (bytes > kSmallCopy && sleepers_.load(std::memory_order_seq_cst) > 0) {
            std::lock_guard<std::mutex> lk(m_);
            cv_work_.notify_all();
        }
        work(g, dst, src, bytes, blk, nblocks, stream_stores);
        // blocks still in other hands: normally < 30 us; a helper that was preempted while holding one can take a scheduler
        // quantum, so after a short spin the caller yields its CPU instead of burning it
        for (unsigned spins = 1; done_.load(std::memory_order_acquire) < nblocks; ++spins) {
            if (spins < 4096) cpu_relax();
            else std::this_thread::yield();
        }
    }
    // Sharded tier: worker pools are sized so that the TOTAL over all devices stays bounded
    // (0 = back to CNT_HOST_COPY_THREADS).  Takes effect at the next copy().
    void set_limit(int n) {
        if (n == limit_) return;
        limit_ = n;
        if (started_) stop();
    }
    // Where the HELPERS run (the caller is never moved).  `domains` = sets of CPUs that share a last-level cache, i.e. on an EPYC
    // one CCD each -- and every CCD reaches memory through a link of its own: four copying threads inside ONE CCD move 59 GB/s
    // together, one thread in each of four CCDs 100 GB/s, eight 152 GB/s (bench/copy_placement_lab.cpp on the GPU box's host,
    // profiles/r06_host_tier.md 2; the scheduler's own placement lands at 84-100).  Helper k is pinned to domain
    // (first + k - 1) mod n: distinct domains as long as there are enough, starting behind the caller's own.  Empty = wherever
    // the caller's mask lets them (CNT_HOST_NUMA=0).  Takes effect at the next copy() (a running team is stopped and restarts).
    void set_domains(const std::vector<cpu_set_t>& domains, size_t first) {
        bool same = domains.size() == domains_.size() && (domains.empty() || first == first_);
        for (size_t i = 0; same && i < domains.size(); ++i) same = CPU_EQUAL(&domains[i], &domains_[i]) != 0;
        if (same) return;
        domains_ = domains;
        first_ = first;
        if (started_) stop();
    }
    int pinned_cpus() const {  // CPUs the helpers may run on, all domains together (0 = not pinned)
        cpu_set_t all;
        CPU_ZERO(&all);
        for (const cpu_set_t& d : domains_) CPU_OR(&all, &all, &d);
        return CPU_COUNT(&all);
    }
    int domains() const { return (int)domains_.size(); }
    // the warm-copy team = what CNT_HOST_COPY_THREADS / the sharded budget count (the caller is one of them); threads that
    // EXIST besides the caller: spawned() -- up to 2 x team - 1, the second half only works on copies into fresh pages
    int size() const { return started_ ? team_ : 0; }
    int spawned() const { return started_ ? (int)workers_.size() : 0; }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_.store(true, std::memory_order_seq_cst);
        }
        cv_work_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
        stop_.store(false, std::memory_order_relaxed);
        started_ = false;
    }

   private:
    // Block sizes (bench/host_fresh_lab.py, profiles/r03_host_copy_blocks.jsonl): warm copies in 1-MiB blocks -- 256 KiB is as
    // fast when the copying threads sit on the staging memory's NUMA node and 25 % slower when they do not (30 vs 23.5 ms
    // per 1-GiB encode); copies into fresh pages in 2-MiB blocks cut on the DESTINATION's 2-MiB grid, so that one thread
    // faults a transparent huge page in instead of eight fighting over it (1-GiB decode into a fresh buffer 35 -> 24 ms).
    static constexpr size_t kMinPar = (size_t)512 << 10, kWarmBlock = (size_t)1 << 20, kFreshBlock = (size_t)2 << 20;
    // (round 6) copies of up to 4 MiB -- the SMALL leg of a pipelined piece is 2 MiB: two 1-MiB blocks occupied two of the team --
    // in 256-KiB blocks: -2...-3 % at 2^26-2^30 nt, -9 % for a 2^22-nt decode (profiles/r06_host_tier.md 13); the 8-MiB copies of the
    // large leg keep 1-MiB blocks, where smaller ones lose 12-25 % (ibid. 4)
    static constexpr size_t kSmallCopy = (size_t)1 << 20, kSmallBlock = (size_t)256 << 10, kSmallBlockMax = (size_t)4 << 20;
    static constexpr int kSpinUs = 150, kStreakUs = 2000;
    // Six since round 6 (four before): with the caller next to the GPU 4, 5, 6 and 8 copy alike; with the caller -- and so its
    // arrays -- on the other socket every copy crosses the socket link once, and 4 threads do not keep it busy (1-GiB decode 24.1-24.3
    // ms against 21.4-22.5 with six, 2^28 nt 6.1-6.3 against 5.4-5.7; profiles/r06_host_tier.md 12)
    static constexpr int kDefaultTeam = 6;
    static size_t small_block_max() {  // CNT_HOST_SMALL_COPY_KI: copies up to this size are cut into 256-KiB blocks (lab knob)
        static const size_t v = [] {
            const char* e = getenv("CNT_HOST_SMALL_COPY_KI");
            const long ki = e ? atol(e) : 0;
            return ki >= 256 && ki <= 16384 ? (size_t)ki << 10 : kSmallBlockMax;
        }();
        return v;
    }
    static size_t warm_block() {  // CNT_HOST_BLOCK_KI: lab knob (bench/host_tier_lab.py blocks)
        static const size_t v = [] {
            const char* e = getenv("CNT_HOST_BLOCK_KI");
            const long ki = e ? atol(e) : 0;
            return ki >= 64 && ki <= 4096 ? (size_t)ki << 10 : kWarmBlock;
        }();
        return v;
    }
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    int threads() {
        if (!started_) {
            started_ = true;
            int t = kDefaultTeam;
            if (const char* e = getenv("CNT_HOST_COPY_THREADS")) t = atoi(e);
            if (limit_ > 0) t = std::min(t, limit_);
            team_ = std::max(1, std::min(t, 16));
            // a team of 1 means what the header says: no helper thread at all, every copy is the caller's memcpy
            n_threads_ = team_ == 1 ? 1 : 2 * team_;  // warm copies use half of them
            const uint64_t seen = gen_.load(std::memory_order_relaxed);
            for (int k = 1; k < n_threads_; ++k) workers_.emplace_back([this, k, seen] { run(k, seen); });
        }
        return n_threads_;
    }
    // take blocks of job `g` until none is left; the generation in the counter's high half keeps a straggler of an
    // older job from ever taking (and losing) a block of this one
    void work(uint64_t g, uint8_t* dst, const uint8_t* src, size_t bytes, size_t blk, uint64_t nblocks, bool stream_stores) {
        const size_t skew = reinterpret_cast<uintptr_t>(dst) & (blk - 1);
        for (;;) {
            uint64_t cur = next_.load(std::memory_order_acquire);
            if ((cur >> 32) != (g & 0xFFFFFFFFull) || (cur & 0xFFFFFFFFull) >= nblocks) return;
            if (!next_.compare_exchange_weak(cur, cur + 1, std::memory_order_acq_rel)) continue;
            const size_t b = (size_t)(cur & 0xFFFFFFFFull);
            const size_t lo = b ? b * blk - skew : 0, hi = std::min(bytes, (b + 1) * blk - skew);
            copy_block(dst + lo, src + lo, hi - lo, stream_stores);
            done_.fetch_add(1, std::memory_order_release);
        }
    }
    void run(int k, uint64_t seen) {
        if (!domains_.empty()) {  // set before the team starts, never changed under it
            const cpu_set_t& mine = domains_[(first_ + (size_t)k - 1) % domains_.size()];
            (void)pthread_setaffinity_np(pthread_self(), sizeof mine, &mine);
        }
        for (;;) {
            // wait for a new generation: spin first (the next copy of a pipelined call is microseconds away), then sleep
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 1; gen_.load(std::memory_order_acquire) == seen && !stop_.load(std::memory_order_relaxed); ++spins) {
                cpu_relax();
                if ((spins & 255u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kSpinUs)) {
                    std::unique_lock<std::mutex> lk(m_);
                    sleepers_.fetch_add(1, std::memory_order_seq_cst);
                    cv_work_.wait(lk, [&] { return stop_.load(std::memory_order_seq_cst) || gen_.load(std::memory_order_seq_cst) != seen; });
                    sleepers_.fetch_sub(1, std::memory_order_seq_cst);
                    break;
                }
            }
            if (stop_.load(std::memory_order_seq_cst)) return;
            const uint64_t g = gen_.load(std::memory_order_acquire);
            if (g == seen) continue;
            seen = g;
            // the job's fields were written before gen_ was and AFTER the block counter moved to their generation (copy()):
            // they all belong to `g`, or at least one belongs to a LATER job -- and then the acquire fence below orders this
            // thread behind that job's bump of the counter, which no longer carries `g`, and work() returns at once
            uint8_t* dst = job_dst_.load(std::memory_order_relaxed);
            const uint8_t* src = job_src_.load(std::memory_order_relaxed);
            const size_t bytes = job_bytes_.load(std::memory_order_relaxed);
            CNT_COPY_POOL_TEST_STALL(k);  // tests/copy_pool_stress.cpp parks a helper HERE, between its field loads, across whole jobs
            const size_t blk = job_blk_.load(std::memory_order_relaxed);
            const uint64_t nblocks = job_nblocks_.load(std::memory_order_relaxed);
            const int team = job_team_.load(std::memory_order_relaxed);
            const bool stream_stores = job_stream_.load(std::memory_order_relaxed);
            std::atomic_thread_fence(std::memory_order_acquire);
            if (k >= team) continue;
            work(g, dst, src, bytes, blk, nblocks, stream_stores);
        }
    }
    std::mutex m_;
    std::condition_variable cv_work_;
    std::vector<std::thread> workers_;
    std::atomic<uint64_t> gen_{0}, next_{0}, done_{0};
    std::atomic<uint8_t*> job_dst_{nullptr};
    std::atomic<const uint8_t*> job_src_{nullptr};
    std::atomic<size_t> job_bytes_{0}, job_blk_{0};
    std::atomic<uint64_t> job_nblocks_{0};
    std::atomic<int> job_team_{0}, sleepers_{0};
    std::atomic<bool> job_stream_{false};
    std::atomic<bool> stop_{false};
    int n_threads_ = 1, team_ = 1, limit_ = 0;
    bool started_ = false;
    std::vector<cpu_set_t> domains_;
    size_t first_ = 0;
    std::chrono::steady_clock::time_point last_mid_copy_{};  // when the calling thread's previous lone mid-size copy ended (a pool has one caller)
};
