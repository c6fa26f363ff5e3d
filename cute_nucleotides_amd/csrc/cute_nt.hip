// cute_nt.hip -- the C-ABI shim of libcute_nt_hip.so (include/cute_nt.h).
//
// Three tiers over the kernels in codec2_kernels.hpp / codec5_kernels.hpp:
//   host-pointer tier   the drop-in for the reference's `&[u8] -> Vec<u64>`
//                       functions: chunked, double-buffered H2D/kernel/D2H;
//   sharded tier        contiguous chunks over the visible devices, one host
//                       thread per device, no collective;
//   device-pointer tier enqueue-only; what the roofline numbers measure.
// No global mutable state apart from the tuning knobs (atomics); streams and
// scratch are per calling thread, so the library is re-entrant like the
// reference's pure functions.
#include "../../include/cute_nt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>

#include "codec2_kernels.hpp"
#include "codec2_launch.hpp"
#include "codec5_kernels.hpp"
#include "codec5_launch.hpp"
#include "util_kernels.hpp"

using namespace cnt;

namespace {

inline int hip_rc(hipError_t e) { return e == hipSuccess ? CNT_OK : -(int)e; }

#define CNT_TRY(expr)                      \
    do {                                   \
        int _rc = (expr);                  \
        if (_rc != CNT_OK) return _rc;     \
    } while (0)
#define HIP_TRY(expr) CNT_TRY(hip_rc(expr))

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---- tuning knobs -------------------------------------------------------------
// Index into kEncodeVariants / kDecodeVariants (codec2_launch.hpp); 0 = shipped default.
std::atomic<int> g_encode_variant{0};
std::atomic<int> g_decode_variant{0};
std::atomic<int> g_encode2_variant{0};
std::atomic<int> g_decode2_variant{0};
// inputs up to this many nucleotides that would need a second (ragged-end) launch anyway go through
// the generic kernel alone: one launch instead of two or three
std::atomic<int> g_small_nt{1 << 17};
std::atomic<int> g_reduce_xi{1};  // hamming / validate tiles take their pages XCD-interleaved (packed_ops_kernels.hpp)
std::atomic<int> g_reduce_fallbacks{0};  // hamming / validate calls that could not get their stream-ordered scratch and ran the generic kernel
std::atomic<int> g_round_trip_shape{0};  // 0 = <64, 4, 1> (default), 1 = <64, 2, 2> (the first shipped shape), codec2_launch.hpp
std::atomic<int> g_round_trip_cap{(int)kRoundTripDefaultCap};  // resident one-wave workgroups per CU of the fused round-trip kernel

inline unsigned generic_grid(uint64_t items) {
    uint64_t b = (items + kBlock - 1) / kBlock;
    return (unsigned)std::min<uint64_t>(std::max<uint64_t>(b, 1), 1u << 16);
}

// ---- per-thread host-copy helpers ---------------------------------------------------
// The host tier stages caller memory through pinned buffers (measured on MI355X / PCIe Gen5,
// profiles/r01_host_tier_lab.log: pageable hipMemcpyAsync runs at 43 GB/s only after the runtime
// has pinned the caller's pages and at 8-14 GB/s on first touch; explicit staging is 20-25 GB/s
// with one copying thread and ~40 GB/s with four, cold or warm).  A tiny fork-join pool does the
// staging copies; CNT_HOST_COPY_THREADS (default 4, 1 = no helpers) sizes it.
class CopyPool {
   public:
    ~CopyPool() { stop(); }
    void copy(uint8_t* dst, const uint8_t* src, size_t bytes) {
        const size_t kMinPar = (size_t)2 << 20;
        if (bytes < kMinPar || threads() <= 1) {
            memcpy(dst, src, bytes);
            return;
        }
        const int T = threads();
        const size_t per = ((bytes + T - 1) / T + 4095) / 4096 * 4096;
        {
            std::lock_guard<std::mutex> lk(m_);
            for (int k = 1; k < T; ++k) {
                const size_t lo = std::min(bytes, per * k), hi = std::min(bytes, per * (k + 1));
                if (hi > lo) {
                    jobs_.push_back({dst + lo, src + lo, hi - lo});
                    ++pending_;
                }
            }
        }
        cv_work_.notify_all();
        memcpy(dst, src, std::min(bytes, per));
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return pending_ == 0; });
    }
    // Sharded tier: worker pools are sized so that the TOTAL over all devices stays bounded
    // (0 = back to CNT_HOST_COPY_THREADS).  Takes effect at the next copy().
    void set_limit(int n) {
        if (n == limit_) return;
        limit_ = n;
        if (started_) stop();
    }
    int size() const { return started_ ? n_threads_ : 0; }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_work_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
        stop_ = false;
        started_ = false;
    }

   private:
    struct Job {
        uint8_t* d;
        const uint8_t* s;
        size_t n;
    };
    int threads() {
        if (!started_) {
            started_ = true;
            int t = 4;
            if (const char* e = getenv("CNT_HOST_COPY_THREADS")) t = atoi(e);
            if (limit_ > 0) t = std::min(t, limit_);
            n_threads_ = std::max(1, std::min(t, 16));
            for (int k = 1; k < n_threads_; ++k) workers_.emplace_back([this] { run(); });
        }
        return n_threads_;
    }
    void run() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
            if (stop_) return;
            Job j = jobs_.back();
            jobs_.pop_back();
            lk.unlock();
            memcpy(j.d, j.s, j.n);
            lk.lock();
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Job> jobs_;
    std::vector<std::thread> workers_;
    size_t pending_ = 0;
    int n_threads_ = 1, limit_ = 0;
    bool stop_ = false, started_ = false;
};

// ---- per-thread device context (streams + grow-only device scratch + pinned staging) ----
struct DevCtx {
    int device = -1;
    hipStream_t stream[2] = {nullptr, nullptr};
    void* d_in[2] = {nullptr, nullptr};
    void* d_out[2] = {nullptr, nullptr};
    uint8_t* h_in[2] = {nullptr, nullptr};   // pinned
    uint8_t* h_out[2] = {nullptr, nullptr};  // pinned
    void* hd_in = nullptr;   // h_in[0] / h_out[0] as the device sees them (zero-copy path for small inputs)
    void* hd_out = nullptr;
    size_t cap_in = 0, cap_out = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};  // device-resident sharded tier: per-shard kernel time

    int ensure_streams() {
        for (int i = 0; i < 2; ++i)
            if (!stream[i]) HIP_TRY(hipStreamCreateWithFlags(&stream[i], hipStreamNonBlocking));
        return CNT_OK;
    }
    int ensure_events() {
        for (int i = 0; i < 2; ++i)
            if (!ev[i]) HIP_TRY(hipEventCreate(&ev[i]));
        return CNT_OK;
    }
    int ensure(size_t need_in, size_t need_out) {
        CNT_TRY(ensure_streams());
        if (need_in > cap_in) {
            cap_in = 0;  // a failure below must not leave a stale capacity behind
            for (int i = 0; i < 2; ++i) {
                if (d_in[i]) HIP_TRY(hipFree(d_in[i]));
                if (h_in[i]) HIP_TRY(hipHostFree(h_in[i]));
                d_in[i] = nullptr;
                h_in[i] = nullptr;
                HIP_TRY(hipMalloc(&d_in[i], need_in));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_in[i]), need_in, hipHostMallocDefault));
            }
            HIP_TRY(hipHostGetDevicePointer(&hd_in, h_in[0], 0));
            cap_in = need_in;
        }
        if (need_out > cap_out) {
            cap_out = 0;
            for (int i = 0; i < 2; ++i) {
                if (d_out[i]) HIP_TRY(hipFree(d_out[i]));
                if (h_out[i]) HIP_TRY(hipHostFree(h_out[i]));
                d_out[i] = nullptr;
                h_out[i] = nullptr;
                HIP_TRY(hipMalloc(&d_out[i], need_out));
                HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h_out[i]), need_out, hipHostMallocDefault));
            }
            HIP_TRY(hipHostGetDevicePointer(&hd_out, h_out[0], 0));
            cap_out = need_out;
        }
        return CNT_OK;
    }
    void release() {
        if (device < 0) return;
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (hipSetDevice(device) == hipSuccess) {
            for (int i = 0; i < 2; ++i) {
                if (stream[i]) (void)hipStreamDestroy(stream[i]);
                if (ev[i]) (void)hipEventDestroy(ev[i]);
                ev[i] = nullptr;
                if (d_in[i]) (void)hipFree(d_in[i]);
                if (d_out[i]) (void)hipFree(d_out[i]);
                if (h_in[i]) (void)hipHostFree(h_in[i]);
                if (h_out[i]) (void)hipHostFree(h_out[i]);
                stream[i] = nullptr;
                d_in[i] = d_out[i] = nullptr;
                h_in[i] = h_out[i] = nullptr;
            }
        }
        cap_in = cap_out = 0;
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

struct ThreadCtx {
    std::map<int, DevCtx> per_device;
    CopyPool pool;
    ~ThreadCtx() {
        pool.stop();
        for (auto& kv : per_device) kv.second.release();
    }
    int get(DevCtx** out) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e == hipErrorNoDevice ? CNT_ENODEV : hip_rc(e);
        DevCtx& c = per_device[dev];
        c.device = dev;
        *out = &c;
        return CNT_OK;
    }
};
thread_local ThreadCtx t_ctx;

// Host-tier chunking: 16 Mi nucleotides per chunk = ~0.4 ms of PCIe time per chunk, 2 x (16 + 16)
// MiB of pinned staging and as much device scratch per calling thread.
constexpr size_t kChunkNt = (size_t)16 << 20;       // multiple of 32
constexpr size_t kChunkNt5 = (size_t)27 * 512 << 10;  // 27-nt words: 512 Ki words per chunk (13.5 Mi nt)

// Small host-tier calls skip the two DMA submissions: the kernels read the pinned staging buffer and
// write the pinned result buffer directly over PCIe (zero-copy), so a call is memcpy, one launch, one
// stream sync, memcpy.  CNT_ZEROCOPY_MAX_NT overrides the size limit (0 disables the path).
size_t zero_copy_max_nt() {
    static const size_t v = [] {
        const char* e = getenv("CNT_ZEROCOPY_MAX_NT");
        return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)1 << 20;
    }();
    return v;
}

// The reference's functions return a FRESH Vec, so the drop-in's copy-out usually lands in pages that have
// never been touched, and the call is page-fault-bound (1 GiB: 11.5 GiB/s with 4 copy threads against 78 GiB/s
// into warm pages on the GPU box's host).  Where transparent huge pages are in `madvise` mode (that host, most
// distributions), advising the 2-MiB-aligned interior of a large output once makes those first touches 2-MiB
// faults: 45.7 GiB/s with the same 4 threads (bench/fresh_pages_lab.cpp, profiles/r02_fresh_pages_lab.log);
// through the library, a 1-GiB decode into a fresh malloc -- what Rust's Vec::with_capacity is -- goes from
// 194 to 72-91 ms, encode from 70 to 35-42 ms (profiles/r02_bench_twin_hugepage_ab.log; 24-25 ms into warm
// pages).  A team of helper threads that ran ahead of the pipeline taking the faults was tried on top and
// bought nothing (same log), so it is not here.  Advice only -- a no-op for memory that is already
// populated, file-backed or under THP=never.  CNT_HOST_HUGEPAGE=0 switches it off.
void advise_huge_output(void* out, size_t bytes) {
    static const bool on = [] {
        const char* e = getenv("CNT_HOST_HUGEPAGE");
        return !(e && e[0] == '0');
    }();
    constexpr uintptr_t kHuge = (uintptr_t)2 << 20;
    if (!on || bytes < 4 * kHuge) return;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(out) + kHuge - 1) & ~(kHuge - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(out) + bytes) & ~(kHuge - 1);
    if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
}

// ---- device-tier bodies (shared by every tier) ------------------------------------
// Alignment plan (profiles/r01_align_lab_*.json): stores that are not 64-B aligned cost ~30 %, loads
// off the 128-B line grid 6-11 %.  Both directions therefore peel a short head through the
// generic kernels until the tile kernels' STORES sit on 128-B lines, and the load side takes
// whatever phase results: the stream kernels when it is zero, otherwise the window kernel
// (encode: line-aligned loads, phase applied to the packed codes) / the funnel-shifting twin
// (decode).  Any pointer the ABI accepts runs at (nearly) full speed; the ragged end goes to the
// generic kernels as before.
int encode_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, hipStream_t s) {
    const size_t words = cnt_words_for(n_len);
    if (out_words < words) return CNT_ECAP;
    if (flags & ~CNT_STRICT_LUT) return CNT_EINVAL;
    if (n_len == 0) return CNT_OK;
    if (!d_n || !d_out || !aligned(d_out, 8)) return CNT_EINVAL;
    const bool strict = (flags & CNT_STRICT_LUT) != 0;
    const uint8_t* n = static_cast<const uint8_t*>(d_n);
    uint64_t* out = static_cast<uint64_t*>(d_out);
    auto generic = [&](uint64_t nt_end, uint64_t first_word, uint64_t end_word) {
        if (first_word >= end_word) return;
        const unsigned g = generic_grid(end_word - first_word);
        if (strict)
            hipLaunchKernelGGL((n_to_bits_generic<true>), dim3(g), dim3(kBlock), 0, s, n, nt_end, out, first_word, end_word);
        else
            hipLaunchKernelGGL((n_to_bits_generic<false>), dim3(g), dim3(kBlock), 0, s, n, nt_end, out, first_word, end_word);
    };
    if (n_len <= (size_t)g_small_nt.load(std::memory_order_relaxed) && (n_len % kWindowEncodeTile) != 0) {
        generic(n_len, 0, words);
        HIP_TRY(hipGetLastError());
        return CNT_OK;
    }
    // head: words until the output is 64-B aligned (enough for the stores; peeling further would only
    // push the input off its own alignment); 256 nt more if the input phase is not zero, so that
    // the window kernel's rounded-down loads stay inside the caller's buffer
    uint64_t head_words = ((64 - (reinterpret_cast<uintptr_t>(d_out) & 63)) & 63) >> 3;
    const uint32_t phase = (uint32_t)((reinterpret_cast<uintptr_t>(n) + 32 * head_words) & 127);
    if (phase && head_words < 4) head_words += 8;  // 64 B of output, 256 B of input: both phases kept
    uint64_t main_nt = 0;
    if (n_len > 32 * head_words) {
        const uint64_t rem = n_len - 32 * head_words;
        const uint8_t* p = n + 32 * head_words;
        uint8_t* o = reinterpret_cast<uint8_t*>(out + head_words);
        const int v = g_encode_variant.load(std::memory_order_relaxed);
        if (phase == 0 || ((phase & 15) == 0 && v != 0)) {  // a non-default variant is honoured whenever it can run
            if (strict ? launch_encode<true>(v, p, o, rem, s, &main_nt) : launch_encode<false>(v, p, o, rem, s, &main_nt))
                return CNT_EINVAL;
        } else if (rem >= kWindowEncodeTile + kWindowEncodeSlack) {
            const uint64_t tiles = (rem - kWindowEncodeSlack) / kWindowEncodeTile;
            if (strict) launch_encode_window<true>(p - phase, phase, o, tiles, s);
            else launch_encode_window<false>(p - phase, phase, o, tiles, s);
            main_nt = tiles * kWindowEncodeTile;
        }
        HIP_TRY(hipGetLastError());
    }
    if (main_nt == 0) {
        generic(n_len, 0, words);
    } else {
        generic(32 * head_words, 0, head_words);
        generic(n_len, head_words + (main_nt >> 5), words);
    }
    HIP_TRY(hipGetLastError());
    return CNT_OK;
}

int decode_dev(const void* d_bits, size_t words, size_t len, void* d_out, unsigned flags, hipStream_t s) {
    if (len > (words << 5) || (words > (SIZE_MAX >> 5))) return CNT_ELEN;
    if (flags) return CNT_EINVAL;
    if (len == 0) return CNT_OK;
    if (!d_bits || !d_out || !aligned(d_bits, 8)) return CNT_EINVAL;
    const size_t used_words = cnt_words_for(len);
    const uint64_t* bits = static_cast<const uint64_t*>(d_bits);
    uint8_t* out = static_cast<uint8_t*>(d_out);
    if (len <= (size_t)g_small_nt.load(std::memory_order_relaxed) && (len % kShiftedDecodeTile) != 0) {
        hipLaunchKernelGGL(bits_to_n_generic, dim3(generic_grid(used_words)), dim3(kBlock), 0, s, bits, (uint64_t)len, out,
                           (uint64_t)0, (uint64_t)used_words);
        HIP_TRY(hipGetLastError());
        return CNT_OK;
    }
    // head: nucleotides until the output is on a 128-B line -- for large buffers on a 4-KiB boundary,
    // so that every tile is one aligned 4-KiB piece (worth 2-5 %)
    const uint64_t grain = len >= ((uint64_t)1 << 20) ? 4096 : 128;
    const uint64_t head = (grain - (reinterpret_cast<uintptr_t>(d_out) & (grain - 1))) & (grain - 1);
    uint64_t main_nt = 0;
    if (len > head) {
        const uint64_t rem = len - head;
        const uint8_t* in = reinterpret_cast<const uint8_t*>(bits) + 4 * (head >> 4);  // dword of nucleotide `head`
        const uint32_t sh = 2 * (uint32_t)(head & 15);
        if (sh == 0) {
            int v = g_decode_variant.load(std::memory_order_relaxed);
            if (!aligned(in, 16) && v == 4) v = 0;  // the lds variant loads 16-B vectors
            if (launch_decode(v, in, out + head, rem, s, &main_nt)) return CNT_EINVAL;
        } else {
            const uint64_t tiles = rem / kShiftedDecodeTile;
            launch_decode_shifted(in, sh, out + head, tiles, s);
            main_nt = tiles * kShiftedDecodeTile;
        }
        HIP_TRY(hipGetLastError());
    }
    if (main_nt == 0) {
        hipLaunchKernelGGL(bits_to_n_generic, dim3(generic_grid(used_words)), dim3(kBlock), 0, s, bits, (uint64_t)len, out,
                           (uint64_t)0, (uint64_t)used_words);
    } else {
        if (head) hipLaunchKernelGGL(bits_to_n_range, dim3(generic_grid(head)), dim3(kBlock), 0, s, bits, out, (uint64_t)0, head);
        const uint64_t lo = head + main_nt;
        if (lo < len) {
            if (lo & 31)
                hipLaunchKernelGGL(bits_to_n_range, dim3(generic_grid(len - lo)), dim3(kBlock), 0, s, bits, out, lo, (uint64_t)len);
            else
                hipLaunchKernelGGL(bits_to_n_generic, dim3(generic_grid(used_words - (lo >> 5))), dim3(kBlock), 0, s, bits,
                                   (uint64_t)len, out, lo >> 5, (uint64_t)used_words);
        }
    }
    HIP_TRY(hipGetLastError());
    return CNT_OK;
}

// 5-letter codec, same alignment plan.  Encode: <= 7 head words make the stores 64-B aligned, the
// input phase goes to n_to_bits2_window.  Decode: the stores are the wide side and 27 is a unit
// mod 128, so a head of k = 19 * (-address mod 128) mod 128 words (27 * 19 = 1 mod 128) puts the
// remaining output on a 128-B line for ANY pointer; the packed side just moves by k words.
// Fused encode + decode (BASELINE.json configs[3]): d_bits = n_to_bits(d_n), d_back = bits_to_n(d_bits,
// n_len), with the ASCII read once and the packed words never read back.  The fused tiles need all
// three pointers on 128-B lines; anything else (and the ragged end) goes through the two ordinary
// entry points, same results.
int round_trip_dev(const void* d_n, size_t n_len, void* d_bits, size_t out_words, void* d_back, unsigned flags, hipStream_t s) {
    const size_t words = cnt_words_for(n_len);
    if (out_words < words) return CNT_ECAP;
    if (flags & ~CNT_STRICT_LUT) return CNT_EINVAL;
    if (n_len == 0) return CNT_OK;
    if (!d_n || !d_bits || !d_back || !aligned(d_bits, 8)) return CNT_EINVAL;
    uint64_t done = 0;
    if (aligned(d_n, 128) && aligned(d_bits, 128) && aligned(d_back, 128)) {
        const uint64_t tiles = n_len / kRoundTripTile;
        const uint32_t cap = (uint32_t)g_round_trip_cap.load(std::memory_order_relaxed);
        const int shape = g_round_trip_shape.load(std::memory_order_relaxed);
        if (flags & CNT_STRICT_LUT)
            launch_round_trip<true>(static_cast<const uint8_t*>(d_n), static_cast<uint8_t*>(d_bits), static_cast<uint8_t*>(d_back), tiles, cap, shape, s);
        else
            launch_round_trip<false>(static_cast<const uint8_t*>(d_n), static_cast<uint8_t*>(d_bits), static_cast<uint8_t*>(d_back), tiles, cap, shape, s);
        HIP_TRY(hipGetLastError());
        done = tiles * kRoundTripTile;  // a multiple of 32: the rest starts on a word
    }
    if (done < n_len) {
        const size_t rest = n_len - done, rest_words = cnt_words_for(rest);
        uint64_t* bits_rest = static_cast<uint64_t*>(d_bits) + (done >> 5);
        CNT_TRY(encode_dev(static_cast<const uint8_t*>(d_n) + done, rest, bits_rest, rest_words, flags, s));
        CNT_TRY(decode_dev(bits_rest, rest_words, rest, static_cast<uint8_t*>(d_back) + done, 0, s));
    }
    return CNT_OK;
}

int encode2_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, hipStream_t s) {
    const size_t words = cnt_words2_for(n_len);
    if (out_words < words) return CNT_ECAP;
    if (flags & ~CNT_STRICT_LUT) return CNT_EINVAL;
    if (n_len == 0) return CNT_OK;
    if (!d_n || !d_out || !aligned(d_out, 8)) return CNT_EINVAL;
    const bool strict = (flags & CNT_STRICT_LUT) != 0;
    const uint8_t* n = static_cast<const uint8_t*>(d_n);
    uint64_t* out = static_cast<uint64_t*>(d_out);
    auto generic = [&](uint64_t nt_end, uint64_t first_word, uint64_t end_word) {
        if (first_word >= end_word) return;
        const unsigned g = generic_grid(end_word - first_word);
        if (strict)
            hipLaunchKernelGGL((n_to_bits2_generic<true>), dim3(g), dim3(kBlock), 0, s, n, nt_end, out, first_word, end_word);
        else
            hipLaunchKernelGGL((n_to_bits2_generic<false>), dim3(g), dim3(kBlock), 0, s, n, nt_end, out, first_word, end_word);
    };
    if (n_len <= (size_t)g_small_nt.load(std::memory_order_relaxed) && (n_len % kWindowEncode2Tile) != 0) {
        generic(n_len, 0, words);
        HIP_TRY(hipGetLastError());
        return CNT_OK;
    }
    uint64_t head_words = ((64 - (reinterpret_cast<uintptr_t>(d_out) & 63)) & 63) >> 3;
    const uint32_t phase = (uint32_t)((reinterpret_cast<uintptr_t>(n) + 27 * head_words) & 127);
    if (phase && head_words < 5) head_words += 8;  // 64 B of output; >= 127 B of input in front of the first window
    const uint32_t phase2 = (uint32_t)((reinterpret_cast<uintptr_t>(n) + 27 * head_words) & 127);
    uint64_t main_words = 0;
    if (n_len > 27 * head_words) {
        const uint64_t rem = n_len - 27 * head_words;
        const uint8_t* p = n + 27 * head_words;
        uint8_t* o = reinterpret_cast<uint8_t*>(out + head_words);
        const int v = g_encode2_variant.load(std::memory_order_relaxed);
        if (phase2 == 0 || ((phase2 & 15) == 0 && v != 0)) {
            if (strict ? launch_encode2<true>(v, p, o, rem, s, &main_words) : launch_encode2<false>(v, p, o, rem, s, &main_words))
                return CNT_EINVAL;
        } else if (rem >= kWindowEncode2Tile + kWindowEncode2Slack) {
            const uint64_t tiles = (rem - kWindowEncode2Slack) / kWindowEncode2Tile;
            if (strict) launch_encode2_window<true>(p - phase2, phase2, o, tiles, s);
            else launch_encode2_window<false>(p - phase2, phase2, o, tiles, s);
            main_words = tiles * (kWindowEncode2Tile / 27);
        }
        HIP_TRY(hipGetLastError());
    }
    if (main_words == 0) {
        generic(n_len, 0, words);
    } else {
        generic(27 * head_words, 0, head_words);
        generic(n_len, head_words + main_words, words);
    }
    HIP_TRY(hipGetLastError());
    return CNT_OK;
}

int decode2_dev(const void* d_bits, size_t words, size_t len, void* d_out, unsigned flags, hipStream_t s) {
    if (words > SIZE_MAX / 27 || len > words * 27) return CNT_ELEN;
    if (flags) return CNT_EINVAL;
    if (len == 0) return CNT_OK;
    if (!d_bits || !d_out || !aligned(d_bits, 8)) return CNT_EINVAL;
    const size_t used_words = cnt_words2_for(len);
    const uint64_t* bits = static_cast<const uint64_t*>(d_bits);
    uint8_t* out = static_cast<uint8_t*>(d_out);
    auto generic = [&](uint64_t nt_end, uint64_t first_word, uint64_t end_word) {
        if (first_word >= end_word) return;
        hipLaunchKernelGGL(bits_to_n2_generic, dim3(generic_grid(end_word - first_word)), dim3(kBlock), 0, s, bits, nt_end, out,
                           first_word, end_word);
    };
    if (len <= (size_t)g_small_nt.load(std::memory_order_relaxed) && (len % kWindowEncode2Tile) != 0) {
        generic(len, 0, used_words);
        HIP_TRY(hipGetLastError());
        return CNT_OK;
    }
    const uint64_t head_words = (19 * ((128 - (reinterpret_cast<uintptr_t>(d_out) & 127)) & 127)) & 127;
    uint64_t main_words = 0;
    if (len > 27 * head_words) {
        if (launch_decode2(g_decode2_variant.load(std::memory_order_relaxed), bits + head_words, out + 27 * head_words,
                           len - 27 * head_words, s, &main_words))
            return CNT_EINVAL;
        HIP_TRY(hipGetLastError());
    }
    if (main_words == 0) {
        generic(len, 0, used_words);
    } else {
        generic(27 * head_words, 0, head_words);
        generic(len, head_words + main_words, used_words);
    }
    HIP_TRY(hipGetLastError());
    return CNT_OK;
}

// ---- host tier: chunked double-buffered pipeline ------------------------------------
// unit_nt: nucleotides per packed word (32 or 27); chunk_nt a multiple of it.
typedef int (*enc_fn)(const void*, size_t, void*, size_t, unsigned, hipStream_t);
typedef int (*dec_fn)(const void*, size_t, size_t, void*, unsigned, hipStream_t);

// Chunk size of the 2-slot pipeline: the full chunk for big inputs; mid-size inputs are cut into
// about four pieces (>= 256 Ki words' worth) so that staging copies and DMA overlap at all.
size_t pipeline_chunk(size_t n_len, size_t unit_nt, size_t chunk_nt) {
    const size_t whole = (n_len + unit_nt - 1) / unit_nt * unit_nt;
    if (whole >= 4 * chunk_nt) return chunk_nt;
    const size_t gran = unit_nt * 8192;  // 256 Ki nt (2-bit) / 216 Ki nt (5-letter)
    size_t c = ((whole / 4 + gran - 1) / gran) * gran;
    if (c < gran) c = gran;
    return std::min(std::min(c, chunk_nt), whole);
}

int host_encode(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, unsigned flags, size_t unit_nt,
                size_t chunk_nt, enc_fn fn) {
    const size_t words = (n_len + unit_nt - 1) / unit_nt;
    if (out_words < words) return CNT_ECAP;
    if (flags & ~CNT_STRICT_LUT) return CNT_EINVAL;
    if (n_len == 0) return CNT_OK;  // empty in -> empty out, no zero-size allocation (SURVEY 8a iv)
    if (!n || !out) return CNT_EINVAL;
    DevCtx* c = nullptr;
    CNT_TRY(t_ctx.get(&c));
    if (n_len <= zero_copy_max_nt()) {
        CNT_TRY(c->ensure(words * unit_nt, words * 8));
        memcpy(c->h_in[0], n, n_len);
        CNT_TRY(fn(c->hd_in, n_len, c->hd_out, words, flags, c->stream[0]));
        HIP_TRY(hipStreamSynchronize(c->stream[0]));
        memcpy(out, c->h_out[0], words * 8);
        return CNT_OK;
    }
    const size_t chunk = pipeline_chunk(n_len, unit_nt, chunk_nt);
    CNT_TRY(c->ensure(chunk, chunk / unit_nt * 8));
    advise_huge_output(out, words * 8);
    // 2-slot pipeline: while slot A's H2D / kernel / D2H run on its stream, the host copies slot B's
    // finished output to the caller and stages slot B's next input.
    size_t pend_word[2] = {0, 0}, pend_words[2] = {0, 0};
    uint8_t* out_bytes = reinterpret_cast<uint8_t*>(out);
    auto retire = [&](int slot) -> int {
        CNT_TRY(hip_rc(hipStreamSynchronize(c->stream[slot])));
        if (pend_words[slot]) t_ctx.pool.copy(out_bytes + pend_word[slot] * 8, c->h_out[slot], pend_words[slot] * 8);
        pend_words[slot] = 0;
        return CNT_OK;
    };
    size_t off = 0;
    int slot = 0, rc = CNT_OK;
    while (off < n_len && rc == CNT_OK) {
        const size_t m = std::min(chunk, n_len - off);
        const size_t w = (m + unit_nt - 1) / unit_nt;
        hipStream_t s = c->stream[slot];
        rc = retire(slot);
        if (rc == CNT_OK) {
            t_ctx.pool.copy(c->h_in[slot], n + off, m);
            rc = hip_rc(hipMemcpyAsync(c->d_in[slot], c->h_in[slot], m, hipMemcpyHostToDevice, s));
        }
        if (rc == CNT_OK) rc = fn(c->d_in[slot], m, c->d_out[slot], w, flags, s);
        if (rc == CNT_OK) rc = hip_rc(hipMemcpyAsync(c->h_out[slot], c->d_out[slot], w * 8, hipMemcpyDeviceToHost, s));
        if (rc == CNT_OK) {
            pend_word[slot] = off / unit_nt;
            pend_words[slot] = w;
        }
        off += m;
        slot ^= 1;
    }
    for (int i = 0; i < 2; ++i, slot ^= 1) {
        int r2 = retire(slot);
        if (rc == CNT_OK) rc = r2;
    }
    return rc;
}

int host_decode(const uint64_t* bits, size_t words, size_t len, uint8_t* out, size_t unit_nt, size_t chunk_nt,
                dec_fn fn) {
    if (words > SIZE_MAX / unit_nt || len > words * unit_nt) return CNT_ELEN;
    if (len == 0) return CNT_OK;
    if (!bits || !out) return CNT_EINVAL;
    DevCtx* c = nullptr;
    CNT_TRY(t_ctx.get(&c));
    if (len <= zero_copy_max_nt()) {
        const size_t w = (len + unit_nt - 1) / unit_nt;
        CNT_TRY(c->ensure(w * 8, len + 32));
        memcpy(c->h_in[0], bits, w * 8);
        CNT_TRY(fn(c->hd_in, w, len, c->hd_out, 0, c->stream[0]));
        HIP_TRY(hipStreamSynchronize(c->stream[0]));
        memcpy(out, c->h_out[0], len);
        return CNT_OK;
    }
    const size_t chunk = pipeline_chunk(len, unit_nt, chunk_nt);
    // +32: the reference's SIMD decoders may store whole 32-B blocks; ours never writes past
    // `len`, the slack only keeps device stores inside the scratch.
    CNT_TRY(c->ensure(chunk / unit_nt * 8, chunk + 32));
    advise_huge_output(out, len);
    size_t pend_off[2] = {0, 0}, pend_n[2] = {0, 0};
    const uint8_t* in_bytes = reinterpret_cast<const uint8_t*>(bits);
    auto retire = [&](int slot) -> int {
        CNT_TRY(hip_rc(hipStreamSynchronize(c->stream[slot])));
        if (pend_n[slot]) t_ctx.pool.copy(out + pend_off[slot], c->h_out[slot], pend_n[slot]);
        pend_n[slot] = 0;
        return CNT_OK;
    };
    size_t off = 0;
    int slot = 0, rc = CNT_OK;
    while (off < len && rc == CNT_OK) {
        const size_t m = std::min(chunk, len - off);
        const size_t w = (m + unit_nt - 1) / unit_nt;
        hipStream_t s = c->stream[slot];
        rc = retire(slot);
        if (rc == CNT_OK) {
            t_ctx.pool.copy(c->h_in[slot], in_bytes + off / unit_nt * 8, w * 8);
            rc = hip_rc(hipMemcpyAsync(c->d_in[slot], c->h_in[slot], w * 8, hipMemcpyHostToDevice, s));
        }
        if (rc == CNT_OK) rc = fn(c->d_in[slot], w, m, c->d_out[slot], 0, s);
        if (rc == CNT_OK) rc = hip_rc(hipMemcpyAsync(c->h_out[slot], c->d_out[slot], m, hipMemcpyDeviceToHost, s));
        if (rc == CNT_OK) {
            pend_off[slot] = off;
            pend_n[slot] = m;
        }
        off += m;
        slot ^= 1;
    }
    for (int i = 0; i < 2; ++i, slot ^= 1) {
        int r2 = retire(slot);
        if (rc == CNT_OK) rc = r2;
    }
    return rc;
}

// ---- sharded tier: one persistent worker thread per shard ------------------------------------
// A worker keeps its device binding and its thread-local context (streams, pinned staging, device
// scratch) between calls; spawning threads per call would re-create 2 x 32 MiB of pinned memory per
// device every time.  The pool is created on first use and deliberately never destroyed: its
// threads sleep on a condition variable until the process exits, so no HIP call can run from a
// static destructor after the runtime is gone.  cnt_shutdown() asks the workers to release their
// contexts.  One sharded call runs at a time (callers queue on call_m_).
//
// Placement on a real node (8 GPUs behind two sockets): the staging copies of a shard are plain
// memcpy between the caller's pages and pinned memory, so the worker -- and the copy-pool threads
// it spawns, which inherit its mask -- is pinned to the CPUs of the NUMA node its GPU hangs off
// (/sys/bus/pci/devices/<bdf>/numa_node, intersected with the mask the process was given;
// CNT_SHARD_NUMA=0 switches it off).  The copy pools are sized so that their TOTAL over all
// devices stays at CNT_SHARD_COPY_THREADS_TOTAL (default 32): 8 GPUs -> 4 threads each, 16 shards
// -> 2 each.
//
// Test hook (never set in production): CNT_SHARD_ALIAS_DEVICES=1 lets ndev exceed the number of
// visible devices (up to kMaxShards) and binds worker k to device k % count, so the partition
// arithmetic, the empty-shard and the ragged-last-shard paths run with ndev > 1 on a 1-GPU box.
constexpr int kMaxShards = 64;

bool shard_alias() {
    const char* e = getenv("CNT_SHARD_ALIAS_DEVICES");
    return e && e[0] == '1';
}

// CPUs of the NUMA node a device sits on (empty set = unknown / not pinning)
struct NumaInfo {
    int node = -1;
    int n_cpus = 0;
    cpu_set_t cpus;
    char bdf[32] = {0};
};

bool parse_cpulist(const char* text, cpu_set_t* out) {  // "0-63,128-191"
    CPU_ZERO(out);
    const char* p = text;
    bool any = false;
    while (*p) {
        char* end = nullptr;
        long lo = strtol(p, &end, 10);
        if (end == p) break;
        long hi = lo;
        p = end;
        if (*p == '-') {
            hi = strtol(p + 1, &end, 10);
            if (end == p + 1) break;
            p = end;
        }
        for (long c = lo; c <= hi && c < CPU_SETSIZE; ++c) {
            if (c >= 0) {
                CPU_SET((int)c, out);
                any = true;
            }
        }
        while (*p == ',' || *p == ' ' || *p == '\n') ++p;
    }
    return any;
}

bool read_small_file(const char* path, char* buf, size_t cap) {
    FILE* f = fopen(path, "r");
    if (!f) return false;
    const size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}

NumaInfo numa_of_device(int device, const cpu_set_t& allowed) {
    NumaInfo info;
    CPU_ZERO(&info.cpus);
    if (hipDeviceGetPCIBusId(info.bdf, (int)sizeof info.bdf, device) != hipSuccess) {
        (void)hipGetLastError();
        info.bdf[0] = 0;
        return info;
    }
    for (char* c = info.bdf; *c; ++c)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');  // sysfs spells the address in lower case
    char path[160], buf[4096];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", info.bdf);
    if (!read_small_file(path, buf, sizeof buf)) return info;
    const int node = atoi(buf);
    if (node < 0) return info;  // -1: the platform does not say
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    cpu_set_t node_cpus;
    if (!read_small_file(path, buf, sizeof buf) || !parse_cpulist(buf, &node_cpus)) return info;
    CPU_AND(&info.cpus, &node_cpus, &allowed);
    info.n_cpus = CPU_COUNT(&info.cpus);
    info.node = node;
    return info;
}

class ShardPool {
   public:
    static ShardPool& get() {
        static ShardPool* p = new ShardPool;  // leaked on purpose, see above
        return *p;
    }
    // fn(k) runs on worker k, bound to device k % count, k in [0, nshards); returns the first non-OK status
    int run(int nshards, int count, const std::function<int(int)>& fn) {
        std::lock_guard<std::mutex> one_call(call_m_);
        std::unique_lock<std::mutex> lk(m_);
        if (workers_.empty()) (void)sched_getaffinity(0, sizeof allowed_, &allowed_);  // the mask the process was given
        while ((int)workers_.size() < nshards) {
            const int k = (int)workers_.size();
            workers_.push_back(new Worker);
            std::thread([this, k] { loop(k); }).detach();
        }
        int total = 32;
        if (const char* e = getenv("CNT_SHARD_COPY_THREADS_TOTAL")) total = atoi(e);
        const int per_worker = std::max(1, total / std::max(1, nshards));
        for (int k = 0; k < nshards; ++k) {
            workers_[k]->job = &fn;
            workers_[k]->want_device = k % count;
            workers_[k]->copy_limit = per_worker;
        }
        pending_ = nshards;
        cv_work_.notify_all();
        cv_done_.wait(lk, [&] { return pending_ == 0; });
        for (int k = 0; k < nshards; ++k)
            if (workers_[k]->rc != CNT_OK) return workers_[k]->rc;
        return CNT_OK;
    }
    int size() {
        std::lock_guard<std::mutex> lk(m_);
        return (int)workers_.size();
    }
    // what worker k is bound to (after its first job): device, NUMA node (-1 unknown), CPUs it may run
    // on, copy-pool threads (0 = its pool has not copied anything big yet)
    int info(int k, int* device, int* node, int* n_cpus, int* copy_threads) {
        std::lock_guard<std::mutex> lk(m_);
        if (k < 0 || k >= (int)workers_.size()) return CNT_EINVAL;
        const Worker& w = *workers_[k];
        if (device) *device = w.device;
        if (node) *node = w.numa.node;
        if (n_cpus) *n_cpus = w.pinned ? w.numa.n_cpus : 0;
        if (copy_threads) *copy_threads = w.copy_threads;
        return CNT_OK;
    }

   private:
    struct Worker {
        const std::function<int(int)>* job = nullptr;
        int rc = CNT_OK;
        int want_device = 0, copy_limit = 0;
        int device = -1, copy_threads = 0;
        bool pinned = false;
        NumaInfo numa;
    };
    void loop(int k) {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            cv_work_.wait(lk, [&] { return workers_[k]->job != nullptr; });
            Worker& w = *workers_[k];
            const std::function<int(int)>* job = w.job;
            const int dev = w.want_device, limit = w.copy_limit;
            const bool rebind = dev != w.device;
            const cpu_set_t allowed = allowed_;
            lk.unlock();
            int rc = hip_rc(hipSetDevice(dev));
            const bool bound = rc == CNT_OK;
            NumaInfo numa;
            bool pinned = false;
            if (bound && rebind) {
                const char* e = getenv("CNT_SHARD_NUMA");
                numa = numa_of_device(dev, allowed);
                if (!(e && e[0] == '0') && numa.n_cpus > 0)
                    pinned = pthread_setaffinity_np(pthread_self(), sizeof numa.cpus, &numa.cpus) == 0;
                else
                    (void)pthread_setaffinity_np(pthread_self(), sizeof allowed, &allowed);
                t_ctx.pool.stop();  // its threads carry the old mask; they restart under the new one
            }
            t_ctx.pool.set_limit(limit);
            if (rc == CNT_OK) rc = (*job)(k);
            const int copy_threads = t_ctx.pool.size();
            lk.lock();
            if (bound && rebind) {
                w.device = dev;
                w.numa = numa;
                w.pinned = pinned;
            }
            w.copy_threads = copy_threads;
            w.rc = rc;
            w.job = nullptr;
            if (--pending_ == 0) cv_done_.notify_all();
        }
    }
    std::mutex call_m_, m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<Worker*> workers_;
    cpu_set_t allowed_;
    int pending_ = 0;
};
std::atomic<bool> g_shard_pool_used{false};

// ndev <= 0: all visible devices.  *count = visible devices (worker k -> device k % count).
int resolve_ndev(int ndev, int* out, int* count_out) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return CNT_ENODEV;
    if (ndev <= 0) ndev = count;
    if (ndev > count && !(shard_alias() && ndev <= kMaxShards)) return CNT_ENODEV;
    *out = ndev;
    *count_out = count;
    return CNT_OK;
}

// The partition: shard k of `ndev` gets nt [k*C, min(N, (k+1)*C)), C = ceil(N/ndev)
// rounded up to a whole number of kernel tiles -- 16384 nt for the 2-bit codec (a multiple of 32),
// 4 x 3456 nt for the 5-letter codec (a multiple of 27) -- so every shard starts on a word (and tile)
// boundary and only the last non-empty shard has a tail; shards past the end are empty (lo == hi == N).
inline size_t shard_gran(int unit_nt) { return unit_nt == 27 ? (size_t)3456 * 4 : (size_t)16384; }
inline void shard_range(size_t n_len, int ndev, int k, int unit_nt, size_t* lo, size_t* hi) {
    const size_t gran = shard_gran(unit_nt);
    size_t per = n_len / ndev + (n_len % ndev ? 1 : 0);
    per = per > SIZE_MAX - gran ? SIZE_MAX : (per / gran + (per % gran ? 1 : 0)) * gran;  // saturate, never wrap
    // k * per cannot overflow for any n_len that exists in memory, but saturate anyway
    const size_t a = (per && (size_t)k > SIZE_MAX / per) ? n_len : std::min(n_len, per * (size_t)k);
    const size_t b = (per && (size_t)(k + 1) > SIZE_MAX / per) ? n_len : std::min(n_len, per * (size_t)(k + 1));
    *lo = a;
    *hi = b;
}

int sharded_host_encode(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev, int unit_nt,
                        int (*fn)(const uint8_t*, size_t, uint64_t*, size_t)) {
    const size_t words = (n_len + unit_nt - 1) / unit_nt;
    if (out_words < words) return CNT_ECAP;
    if (n_len == 0) return CNT_OK;
    if (!n || !out) return CNT_EINVAL;
    int count = 0;
    CNT_TRY(resolve_ndev(ndev, &ndev, &count));
    g_shard_pool_used.store(true);
    return ShardPool::get().run(ndev, count, [=](int k) -> int {
        size_t lo, hi;
        shard_range(n_len, ndev, k, unit_nt, &lo, &hi);
        if (lo >= hi) return CNT_OK;
        return fn(n + lo, hi - lo, out + lo / unit_nt, (hi - lo + unit_nt - 1) / unit_nt);
    });
}

int sharded_host_decode(const uint64_t* bits, size_t words, size_t len, uint8_t* out, int ndev, int unit_nt,
                        int (*fn)(const uint64_t*, size_t, size_t, uint8_t*)) {
    if (words > SIZE_MAX / unit_nt || len > words * unit_nt) return CNT_ELEN;
    if (len == 0) return CNT_OK;
    if (!bits || !out) return CNT_EINVAL;
    int count = 0;
    CNT_TRY(resolve_ndev(ndev, &ndev, &count));
    g_shard_pool_used.store(true);
    return ShardPool::get().run(ndev, count, [=](int k) -> int {
        size_t lo, hi;
        shard_range(len, ndev, k, unit_nt, &lo, &hi);
        if (lo >= hi) return CNT_OK;
        return fn(bits + lo / unit_nt, (hi - lo + unit_nt - 1) / unit_nt, hi - lo, out + lo);
    });
}

// Device-resident sharded tier: shard k already lives on device k % count; the calling thread
// enqueues every shard on that device's own stream (enqueue is asynchronous, so the devices run
// concurrently without helper threads), then waits for all of them.  No host staging, no collective.
template <typename Enqueue>
int sharded_dev_run(int ndev, float* shard_ms, Enqueue&& enqueue) {
    int count = 0;
    CNT_TRY(resolve_ndev(ndev, &ndev, &count));
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    int rc = CNT_OK;
    std::vector<DevCtx*> ctx((size_t)ndev, nullptr);
    std::vector<int> slot((size_t)ndev, 0);
    for (int k = 0; k < ndev && rc == CNT_OK; ++k) {
        rc = hip_rc(hipSetDevice(k % count));
        if (rc == CNT_OK) rc = t_ctx.get(&ctx[k]);
        if (rc == CNT_OK) rc = ctx[k]->ensure_streams();
        if (rc != CNT_OK) {
            ctx[k] = nullptr;
            break;
        }
        // aliased shards (test hook) share a device: they alternate between its two streams; timing needs
        // the device's single event pair to itself, so with events the aliased shards run one after the other
        slot[k] = (k / count) & 1;
        hipStream_t s = ctx[k]->stream[slot[k]];
        if (shard_ms) {
            rc = ctx[k]->ensure_events();
            if (rc == CNT_OK && k >= count) rc = hip_rc(hipStreamSynchronize(ctx[k - count]->stream[slot[k - count]]));
            if (rc == CNT_OK && k >= count) {
                float ms = 0.f;
                rc = hip_rc(hipEventElapsedTime(&ms, ctx[k]->ev[0], ctx[k]->ev[1]));
                shard_ms[k - count] = ms;
            }
            if (rc == CNT_OK) rc = hip_rc(hipEventRecord(ctx[k]->ev[0], s));
        }
        if (rc == CNT_OK) rc = enqueue(k, s);
        if (rc == CNT_OK && shard_ms) rc = hip_rc(hipEventRecord(ctx[k]->ev[1], s));
    }
    for (int k = 0; k < ndev; ++k) {  // wait for everything that was enqueued, also after an error
        if (!ctx[k]) continue;
        int r2 = hip_rc(hipSetDevice(k % count));
        if (r2 == CNT_OK) r2 = hip_rc(hipStreamSynchronize(ctx[k]->stream[slot[k]]));
        if (r2 == CNT_OK && shard_ms && rc == CNT_OK && k + count >= ndev) {
            float ms = 0.f;
            r2 = hip_rc(hipEventElapsedTime(&ms, ctx[k]->ev[0], ctx[k]->ev[1]));
            shard_ms[k] = ms;
        }
        if (rc == CNT_OK) rc = r2;
    }
    (void)hipSetDevice(prev);
    return rc;
}

}  // namespace

// =====================================================================================
// exported C ABI
// =====================================================================================
extern "C" {

const char* cnt_strerror(int status) {
    switch (status) {
        case CNT_OK: return "ok";
        case CNT_EINVAL: return "invalid argument (null pointer with non-zero size, unknown flag, bad alignment of a word pointer)";
        case CNT_ECAP: return "output capacity too small";
        case CNT_ELEN: return "The length is greater than the number of nucleotides!";  // n_to_bits.rs:53
        case CNT_ENODEV: return "no such HIP device";
        case CNT_ERANGE: return "shard offset is not on a word boundary";
        default: break;
    }
    if (status < 0) return hipGetErrorString((hipError_t)(-status));
    return "unknown status";
}

int cnt_abi_version(void) { return CNT_ABI_VERSION; }

size_t cnt_words_for(size_t n_len) { return (n_len >> 5) + ((n_len & 31) ? 1 : 0); }
size_t cnt_words2_for(size_t n_len) { return n_len / 27 + ((n_len % 27) ? 1 : 0); }

int cnt_device_count(int* count) {
    if (!count) return CNT_EINVAL;
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) {
        *count = 0;
        return CNT_OK;
    }
    return hip_rc(e);
}

int cnt_set_device(int device) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CNT_ENODEV;
    return hip_rc(hipSetDevice(device));
}

int cnt_get_device(int* device) {
    if (!device) return CNT_EINVAL;
    return hip_rc(hipGetDevice(device));
}

// identity / placement of a device, for multi-GPU reports and for callers that pin their own threads
int cnt_device_pci_bus_id(int device, char* buf, size_t cap) {
    if (!buf || cap < 13) return CNT_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CNT_ENODEV;
    return hip_rc(hipDeviceGetPCIBusId(buf, (int)std::min<size_t>(cap, 64), device));
}

int cnt_device_numa_node(int device, int* node) {
    if (!node) return CNT_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CNT_ENODEV;
    cpu_set_t all;
    CPU_ZERO(&all);
    for (int c = 0; c < CPU_SETSIZE; ++c) CPU_SET(c, &all);
    *node = numa_of_device(device, all).node;
    return CNT_OK;
}

static int release_thread_ctx() {
    t_ctx.pool.stop();
    for (auto& kv : t_ctx.per_device) kv.second.release();
    t_ctx.per_device.clear();
    return CNT_OK;
}

int cnt_shutdown(void) {
    release_thread_ctx();
    if (g_shard_pool_used.load()) {  // the sharded tier's workers hold contexts of their own
        ShardPool& pool = ShardPool::get();
        const int n = pool.size();
        int count = 0;
        if (n > 0 && hipGetDeviceCount(&count) == hipSuccess && count > 0)
            return pool.run(n, count, [](int) -> int { return release_thread_ctx(); });
    }
    return CNT_OK;
}

// ---- host tier ----------------------------------------------------------------------
int cnt_n_to_bits_ex(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, unsigned flags) {
    return host_encode(n, n_len, out, out_words, flags, 32, kChunkNt, encode_dev);
}
int cnt_n_to_bits(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words) {
    return cnt_n_to_bits_ex(n, n_len, out, out_words, 0);
}
int cnt_bits_to_n(const uint64_t* bits, size_t words, size_t len, uint8_t* out) {
    return host_decode(bits, words, len, out, 32, kChunkNt, decode_dev);
}
int cnt_n_to_bits2(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words) {
    return host_encode(n, n_len, out, out_words, 0, 27, kChunkNt5, encode2_dev);
}
int cnt_bits_to_n2(const uint64_t* bits, size_t words, size_t len, uint8_t* out) {
    return host_decode(bits, words, len, out, 27, kChunkNt5, decode2_dev);
}

// ---- sharded tier -------------------------------------------------------------------
// Partition: shard_range() above.  No collective: outputs are disjoint ranges of `out`.
int cnt_n_to_bits_sharded(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev) {
    return sharded_host_encode(n, n_len, out, out_words, ndev, 32, cnt_n_to_bits);
}
int cnt_bits_to_n_sharded(const uint64_t* bits, size_t words, size_t len, uint8_t* out, int ndev) {
    return sharded_host_decode(bits, words, len, out, ndev, 32, cnt_bits_to_n);
}
// 5-letter codec over N GPUs: same scheme, shards are whole numbers of 128-word tiles (3456 nt)
int cnt_n_to_bits2_sharded(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev) {
    return sharded_host_encode(n, n_len, out, out_words, ndev, 27, cnt_n_to_bits2);
}
int cnt_bits_to_n2_sharded(const uint64_t* bits, size_t words, size_t len, uint8_t* out, int ndev) {
    return sharded_host_decode(bits, words, len, out, ndev, 27, cnt_bits_to_n2);
}

int cnt_shard_range(size_t n_len, int ndev, int k, int nt_per_word, size_t* lo, size_t* hi) {
    if (!lo || !hi || ndev <= 0 || k < 0 || k >= ndev || (nt_per_word != 32 && nt_per_word != 27)) return CNT_EINVAL;
    shard_range(n_len, ndev, k, nt_per_word, lo, hi);
    return CNT_OK;
}

int cnt_shard_worker_info(int k, int* device, int* numa_node, int* n_cpus, int* copy_threads) {
    if (!g_shard_pool_used.load()) return CNT_EINVAL;
    return ShardPool::get().info(k, device, numa_node, n_cpus, copy_threads);
}

// device-resident shards: one entry per shard, shard k on device k (k % count under the test hook)
static int sharded_dev_encode(const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, int ndev,
                              unsigned flags, float* shard_ms, enc_fn fn) {
    if (!d_n || !n_len || !d_out || !out_words) return CNT_EINVAL;
    return sharded_dev_run(ndev, shard_ms, [&](int k, hipStream_t s) { return fn(d_n[k], n_len[k], d_out[k], out_words[k], flags, s); });
}
static int sharded_dev_decode(const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, int ndev,
                              unsigned flags, float* shard_ms, dec_fn fn) {
    if (!d_bits || !words || !len || !d_out) return CNT_EINVAL;
    return sharded_dev_run(ndev, shard_ms, [&](int k, hipStream_t s) { return fn(d_bits[k], words[k], len[k], d_out[k], flags, s); });
}
int cnt_n_to_bits_sharded_dev(const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, int ndev,
                              unsigned flags, float* shard_ms) {
    return sharded_dev_encode(d_n, n_len, d_out, out_words, ndev, flags, shard_ms, encode_dev);
}
int cnt_bits_to_n_sharded_dev(const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, int ndev,
                              unsigned flags, float* shard_ms) {
    return sharded_dev_decode(d_bits, words, len, d_out, ndev, flags, shard_ms, decode_dev);
}
int cnt_n_to_bits2_sharded_dev(const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, int ndev,
                               unsigned flags, float* shard_ms) {
    return sharded_dev_encode(d_n, n_len, d_out, out_words, ndev, flags, shard_ms, encode2_dev);
}
int cnt_bits_to_n2_sharded_dev(const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, int ndev,
                               unsigned flags, float* shard_ms) {
    return sharded_dev_decode(d_bits, words, len, d_out, ndev, flags, shard_ms, decode2_dev);
}

// ---- device tier --------------------------------------------------------------------
int cnt_n_to_bits_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, void* stream) {
    return encode_dev(d_n, n_len, d_out, out_words, flags, static_cast<hipStream_t>(stream));
}
int cnt_bits_to_n_dev(const void* d_bits, size_t words, size_t len, void* d_out, unsigned flags, void* stream) {
    return decode_dev(d_bits, words, len, d_out, flags, static_cast<hipStream_t>(stream));
}
int cnt_round_trip_dev(const void* d_n, size_t n_len, void* d_bits, size_t out_words, void* d_back, unsigned flags, void* stream) {
    return round_trip_dev(d_n, n_len, d_bits, out_words, d_back, flags, static_cast<hipStream_t>(stream));
}
int cnt_n_to_bits2_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, void* stream) {
    return encode2_dev(d_n, n_len, d_out, out_words, flags, static_cast<hipStream_t>(stream));
}
int cnt_bits_to_n2_dev(const void* d_bits, size_t words, size_t len, void* d_out, unsigned flags, void* stream) {
    return decode2_dev(d_bits, words, len, d_out, flags, static_cast<hipStream_t>(stream));
}

// ---- device memory for callers that do not link HIP themselves (Rust / C / C++ benches) ----------
int cnt_dev_alloc(void** d_ptr, size_t bytes) {
    if (!d_ptr) return CNT_EINVAL;
    *d_ptr = nullptr;
    if (bytes == 0) return CNT_OK;
    return hip_rc(hipMalloc(d_ptr, bytes));
}
int cnt_dev_free(void* d_ptr) { return d_ptr ? hip_rc(hipFree(d_ptr)) : CNT_OK; }
int cnt_dev_upload(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return CNT_OK;
    if (!d_dst || !h_src) return CNT_EINVAL;
    return hip_rc(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
}
int cnt_dev_download(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return CNT_OK;
    if (!h_dst || !d_src) return CNT_EINVAL;
    return hip_rc(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
}
int cnt_dev_sync(void* stream) { return hip_rc(hipStreamSynchronize(static_cast<hipStream_t>(stream))); }

// ---- utilities ----------------------------------------------------------------------
int cnt_fill_random_acgt_dev(void* d_out, size_t first_nt, size_t n_len, uint64_t seed, void* stream) {
    if (first_nt & 31) return CNT_ERANGE;
    if (n_len == 0) return CNT_OK;
    if (!d_out) return CNT_EINVAL;
    const uint64_t blocks = (n_len + 31) >> 5;
    hipLaunchKernelGGL(fill_random_acgt, dim3(generic_grid(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint8_t*>(d_out), (uint64_t)(first_nt >> 5), (uint64_t)n_len, seed,
                       aligned(d_out, 16) ? 1 : 0);
    return hip_rc(hipGetLastError());
}

int cnt_fill_random_acgtn_dev(void* d_out, size_t first_nt, size_t n_len, uint64_t seed, void* stream) {
    if (first_nt % 27) return CNT_ERANGE;
    if (n_len == 0) return CNT_OK;
    if (!d_out) return CNT_EINVAL;
    const uint64_t blocks = (n_len + 26) / 27;
    hipLaunchKernelGGL(fill_random_acgtn, dim3(generic_grid(blocks)), dim3(kBlock), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint8_t*>(d_out), (uint64_t)(first_nt / 27), (uint64_t)n_len, seed);
    return hip_rc(hipGetLastError());
}

int cnt_checksum_words_dev(const void* d_words, size_t first_word, size_t words, void* d_sum, void* stream) {
    if (words == 0) return CNT_OK;
    if (!d_words || !d_sum || !aligned(d_words, 8) || !aligned(d_sum, 8)) return CNT_EINVAL;
    hipLaunchKernelGGL(checksum_words, dim3(std::min(generic_grid(words), 4096u)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint64_t*>(d_words), (uint64_t)first_word,
                       (uint64_t)words, static_cast<unsigned long long*>(d_sum));
    return hip_rc(hipGetLastError());
}

int cnt_count_mismatch_dev(const void* d_a, const void* d_b, size_t nbytes, void* d_count, void* stream) {
    if (nbytes == 0) return CNT_OK;
    if (!d_a || !d_b || !d_count || !aligned(d_count, 8)) return CNT_EINVAL;
    hipLaunchKernelGGL(count_mismatch, dim3(std::min(generic_grid(nbytes >> 4), 4096u)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint8_t*>(d_a), static_cast<const uint8_t*>(d_b),
                       (uint64_t)nbytes, (aligned(d_a, 16) && aligned(d_b, 16)) ? 1 : 0,
                       static_cast<unsigned long long*>(d_count));
    return hip_rc(hipGetLastError());
}

// ---- tuning -------------------------------------------------------------------------
int cnt_set_tuning(const char* key, int value) {
    if (!key) return CNT_EINVAL;
    if (!strcmp(key, "encode")) {
        if (value < 0 || value >= kNumEncodeVariants) return CNT_EINVAL;
        g_encode_variant.store(value);
    } else if (!strcmp(key, "decode")) {
        if (value < 0 || value >= kNumDecodeVariants) return CNT_EINVAL;
        g_decode_variant.store(value);
    } else if (!strcmp(key, "encode2")) {
        if (value < 0 || value >= kNumEncode2Variants) return CNT_EINVAL;
        g_encode2_variant.store(value);
    } else if (!strcmp(key, "decode2")) {
        if (value < 0 || value >= kNumDecode2Variants) return CNT_EINVAL;
        g_decode2_variant.store(value);
    } else if (!strcmp(key, "round_trip_cap")) {
        if (value < 0 || value > 32) return CNT_EINVAL;
        g_round_trip_cap.store(value);
    } else if (!strcmp(key, "round_trip_shape")) {
        if (value < 0 || value > 1) return CNT_EINVAL;
        g_round_trip_shape.store(value);
    } else if (!strcmp(key, "reduce_xi")) {
        if (value < 0 || value > 1) return CNT_EINVAL;
        g_reduce_xi.store(value);
    } else if (!strcmp(key, "small_nt")) {
        if (value < 0) return CNT_EINVAL;
        g_small_nt.store(value);
    } else {
        return CNT_EINVAL;
    }
    return CNT_OK;
}

int cnt_get_tuning(const char* key, int* value) {
    if (!key || !value) return CNT_EINVAL;
    if (!strcmp(key, "encode")) *value = g_encode_variant.load();
    else if (!strcmp(key, "decode")) *value = g_decode_variant.load();
    else if (!strcmp(key, "encode2")) *value = g_encode2_variant.load();
    else if (!strcmp(key, "decode2")) *value = g_decode2_variant.load();
    else if (!strcmp(key, "small_nt")) *value = g_small_nt.load();
    else if (!strcmp(key, "reduce_xi")) *value = g_reduce_xi.load();
    else if (!strcmp(key, "round_trip_cap")) *value = g_round_trip_cap.load();
    else if (!strcmp(key, "round_trip_shape")) *value = g_round_trip_shape.load();
    else if (!strcmp(key, "reduce_fallbacks")) *value = g_reduce_fallbacks.load();
    else if (!strcmp(key, "encode_variants")) *value = kNumEncodeVariants;
    else if (!strcmp(key, "decode_variants")) *value = kNumDecodeVariants;
    else if (!strcmp(key, "encode2_variants")) *value = kNumEncode2Variants;
    else if (!strcmp(key, "decode2_variants")) *value = kNumDecode2Variants;
    else return CNT_EINVAL;
    return CNT_OK;
}

const char* cnt_tuning_name(const char* key, int value) {
    if (!key) return nullptr;
    if (!strcmp(key, "encode") && value >= 0 && value < kNumEncodeVariants) return kEncodeVariants[value].name;
    if (!strcmp(key, "decode") && value >= 0 && value < kNumDecodeVariants) return kDecodeVariants[value].name;
    if (!strcmp(key, "encode2") && value >= 0 && value < kNumEncode2Variants) return kEncode2Variants[value].name;
    if (!strcmp(key, "decode2") && value >= 0 && value < kNumDecode2Variants) return kDecode2Variants[value].name;
    return nullptr;
}

}  // extern "C"

#include "packed_ops_abi.inc"
