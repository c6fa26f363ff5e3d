// cute_nt.hip -- the C-ABI shim of libcute_nt_hip.so (include/cute_nt.h).
//
// Three tiers over the kernels in codec2_kernels.hpp / codec5_kernels.hpp:
//   host-pointer tier   the drop-in for the reference's `&[u8] -> Vec<u64>`
//                       functions: chunked, double-buffered H2D/kernel/D2H;
//   sharded tier        contiguous chunks over the visible devices, one host
//                       thread per device, no collective;
//   device-pointer tier enqueue-only; what the roofline numbers measure.
// No global mutable state that selects code (the tuning knobs exist in the lab build only, see below); streams and
// scratch are per calling thread, so the library is re-entrant like the reference's pure functions.
//
// One translation unit (the kernel headers define non-template __global__ functions), split for reading:
//   shim_host_ctx.inc   staging-copy pool, per-thread / per-device context, huge-page advice
//   device_tier.inc     alignment plan + kernel selection: encode_dev, decode_dev, round_trip_dev, *2_dev
//   host_tier.inc       zero-copy small calls, pinned-staging pipeline over a ring of 3 (2..4) slots
//   sharded_tier.inc    partition, NUMA-pinned worker pool, resident-shard runner
//   (this file)         tuning knobs and every exported symbol of include/cute_nt.h
//   packed_ops_abi.inc  the packed-domain operations' entry points
#include "../include/cute_nt.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
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
#include <sys/syscall.h>
#include <unistd.h>

#include "codec2_kernels.hpp"
#include "codec2_launch.hpp"
#include "codec5_kernels.hpp"
#include "codec5_launch.hpp"
#include "util_kernels.hpp"

using namespace cnt;

namespace {

// A failed runtime call leaves its code in the thread's "last error", which the NEXT hipGetLastError() -- the check behind every
// kernel launch here -- would report as that launch's: a status handed to the caller is cleared from the thread first.
inline int hip_rc(hipError_t e) {
    if (e == hipSuccess) return CNT_OK;
    (void)hipGetLastError();
    return -(int)e;
}

#define CNT_TRY(expr)                      \
    do {                                   \
        int _rc = (expr);                  \
        if (_rc != CNT_OK) return _rc;     \
    } while (0)
#define HIP_TRY(expr) CNT_TRY(hip_rc(expr))

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }
// The reference borrows its input and returns a fresh Vec (n_to_bits.rs:34,51): input and output never share memory.  With
// caller-owned outputs they could; every codec entry point refuses that (CNT_EINVAL) instead of racing its tiles' reads
// against other tiles' stores.
inline bool overlaps(const void* a, size_t a_bytes, const void* b, size_t b_bytes) {
    const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
    return a_bytes && b_bytes && x < y + b_bytes && y < x + a_bytes;
}

// ---- tuning knobs: LAB BUILD ONLY -------------------------------------------------------
// The product library selects nothing at run time: every tune_*() below is a compile-time constant there, the variant
// tables hold their default entry only, and cnt_set_tuning() answers CNT_EINVAL for every key -- no thread can change the
// kernel under another thread's call (the header's re-entrancy contract; the reference's functions are pure over
// immutable tables, n_to_bits.rs:8,23).  -DCNT_LAB_VARIANTS (bench/libcute_nt_hip_lab.so) turns them into process-global
// atomics for A/B runs and for the tests that walk variants, tile maps and the several-launch loops.
#ifdef CNT_LAB_VARIANTS
// Index into kEncodeVariants / kDecodeVariants (codec2_launch.hpp); 0 = shipped default.
std::atomic<int> g_encode_variant{0};
std::atomic<int> g_decode_variant{0};
std::atomic<int> g_encode2_variant{0};
std::atomic<int> g_decode2_variant{0};
// inputs up to this many nucleotides that would need a second (ragged-end) launch anyway go through
// the generic kernel alone: one launch instead of two or three
std::atomic<int> g_small_nt{1 << 17};
std::atomic<int> g_reduce_xi{1};  // hamming / validate tiles take their pages XCD-interleaved (packed_ops_kernels.hpp)
std::atomic<int> g_reduce_persistent{1};  // 1 = hamming / validate as one launch of persistent waves (default); 0 = tiles + scratch + second pass (round 1)
std::atomic<int> g_hamming_order{0};  // load order of hamming_persist's two streams (packed_ops_kernels.hpp): 0 interleaved (shipped), 1 blocks, 2 skewed
std::atomic<int> g_reduce_fallbacks{0};  // hamming / validate calls that could not get their stream-ordered scratch and ran the generic kernel
std::atomic<int> g_round_trip_shape{0};  // 0 = <64, 4, 1> (default), 1 = <64, 2, 2> (the first shipped shape), codec2_launch.hpp
std::atomic<int> g_round_trip_cap{(int)kRoundTripDefaultCap};  // resident one-wave workgroups per CU of the fused round-trip kernel
inline int tune_encode() { return g_encode_variant.load(std::memory_order_relaxed); }
inline int tune_decode() { return g_decode_variant.load(std::memory_order_relaxed); }
inline int tune_encode2() { return g_encode2_variant.load(std::memory_order_relaxed); }
inline int tune_decode2() { return g_decode2_variant.load(std::memory_order_relaxed); }
inline size_t tune_small_nt() { return (size_t)g_small_nt.load(std::memory_order_relaxed); }
inline int tune_round_trip_shape() { return g_round_trip_shape.load(std::memory_order_relaxed); }
inline uint32_t tune_round_trip_cap() { return (uint32_t)g_round_trip_cap.load(std::memory_order_relaxed); }
std::atomic<int> g_round_trip_window_map{0};  // tile map of the any-alignment fused kernel: 0 plain order (shipped), 1 XCD pairs, 2 XCD quads
inline int tune_round_trip_window_map() { return g_round_trip_window_map.load(std::memory_order_relaxed); }
std::atomic<int> g_round_trip_plan{kRoundTripDefaultPlan};  // pricing of the any-alignment fused launch plan (device_tier.inc round_trip_plan): 3 = shipped; 0 tiles on d_back's pages, 1 windows on d_n's pages, 2 shortest read-ahead
inline int tune_round_trip_plan() { return g_round_trip_plan.load(std::memory_order_relaxed); }
std::atomic<int> g_decode_rot{-1};  // further 4-KiB output pages peeled in front of decode's tiles (device_tier.inc decode_turn_pages): -1 the shipped rule, 0..3 forced, 10 / 11 the two candidate rules
inline int tune_decode_rot() { return g_decode_rot.load(std::memory_order_relaxed); }
std::atomic<int> g_decode_window{1};  // 1 (shipped): bits_to_n_window for calls past the Infinity Cache whose packed stream is off its lines or dwords; 0: round 4's stream / shifted kernels
inline bool tune_decode_window() { return g_decode_window.load(std::memory_order_relaxed) != 0; }
// log2 of the size (nt) beyond which decode plans for a packed stream that no longer fits the Infinity Cache: -1 = the device's
// (chip_info().cache_nt, the shipped rule), 0..40 forced -- 0 lets a 2^20-nt call walk the past-the-cache plan (turn placement +
// bits_to_n_window) at every packed phase with guard pages around it (its only coverage used to need 4 GiB of HBM)
std::atomic<int> g_decode_cache_log2{-1};
inline uint64_t decode_cache_nt() {
    const int o = g_decode_cache_log2.load(std::memory_order_relaxed);
    return o >= 0 ? (uint64_t)1 << o : chip_info().cache_nt;
}
#else
constexpr int tune_encode() { return 0; }
constexpr int tune_decode() { return 0; }
constexpr int tune_encode2() { return 0; }
constexpr int tune_decode2() { return 0; }
constexpr size_t tune_small_nt() { return (size_t)1 << 17; }
constexpr int tune_round_trip_shape() { return 0; }
constexpr uint32_t tune_round_trip_cap() { return kRoundTripDefaultCap; }
constexpr int tune_round_trip_window_map() { return 0; }
constexpr int tune_round_trip_plan() { return kRoundTripDefaultPlan; }
constexpr int tune_decode_rot() { return -1; }
constexpr bool tune_decode_window() { return true; }
inline uint64_t decode_cache_nt() { return chip_info().cache_nt; }
#endif

inline unsigned generic_grid(uint64_t items) {
    uint64_t b = (items + kBlock - 1) / kBlock;
    return (unsigned)std::min<uint64_t>(std::max<uint64_t>(b, 1), 1u << 16);
}

#include "shim_host_ctx.inc"
#include "device_tier.inc"
#include "host_tier.inc"
#include "sharded_tier.inc"

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

// what the calling thread's host tier is bound to on its current device (after its first host-tier call there)
int cnt_host_tier_info(int* device, int* numa_node, int* helper_cpus, int* staging_node) {
    DevCtx* c = nullptr;
    CNT_TRY(t_ctx.get(&c));
    if (device) *device = c->device;
    if (numa_node) *numa_node = c->numa.node;
    if (helper_cpus) *helper_cpus = t_ctx.pool.pinned_cpus();
    if (staging_node) *staging_node = numa_node_of_page(c->h_in[0]);
    return CNT_OK;
}

// ---- pinned caller memory: the host tier's fast lane ----------------------------------------------------------------
// Nothing here keeps a table: whether a slice is pinned is asked of the runtime at every call (host_range_is_pinned), so a
// buffer freed or unregistered behind the library's back is simply staged again.
int cnt_host_alloc(void** p, size_t bytes) {
    if (!p || !bytes) return CNT_EINVAL;
    *p = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return CNT_ENODEV;
    }
    HIP_TRY(hipHostMalloc(p, bytes, hipHostMallocPortable | hipHostMallocMapped));
    return CNT_OK;
}
int cnt_host_free(void* p) {
    if (!p) return CNT_OK;
    HIP_TRY(hipHostFree(p));
    return CNT_OK;
}
int cnt_host_register(void* p, size_t bytes) {
    if (!p || !bytes) return CNT_EINVAL;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return CNT_ENODEV;
    }
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterPortable | hipHostRegisterMapped));
    return CNT_OK;
}
int cnt_host_unregister(void* p) {
    if (!p) return CNT_EINVAL;
    HIP_TRY(hipHostUnregister(p));
    return CNT_OK;
}
int cnt_host_is_pinned(const void* p, size_t bytes) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    return host_range_is_pinned(p, bytes) ? 1 : 0;
}

static int release_thread_ctx() {
    t_queues.release();
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
    return host_encode(n, n_len, out, out_words, flags, lut_from_of(n_len, flags, 32), 32, kChunkNt, encode_impl);
}
int cnt_n_to_bits(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words) {
    return cnt_n_to_bits_ex(n, n_len, out, out_words, 0);
}
// the same call, also counting the bytes outside ACGTUacgtu -- in the encode kernels' own pass over the data
int cnt_n_to_bits_checked(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, unsigned flags, uint64_t* invalid) {
    if (!invalid) return CNT_EINVAL;
    return host_encode(n, n_len, out, out_words, flags, lut_from_of(n_len, flags, 32), 32, kChunkNt, encode_impl, invalid);
}
int cnt_bits_to_n(const uint64_t* bits, size_t words, size_t len, uint8_t* out) {
    return host_decode(bits, words, len, out, 32, kChunkNt, decode_dev);
}
int cnt_n_to_bits2_ex(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, unsigned flags) {
    return host_encode(n, n_len, out, out_words, flags, lut_from_of(n_len, flags, 27), 27, kChunkNt5, encode2_impl);
}
int cnt_n_to_bits2(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words) {
    return cnt_n_to_bits2_ex(n, n_len, out, out_words, 0);
}
int cnt_n_to_bits2_checked(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, unsigned flags, uint64_t* invalid) {
    if (!invalid) return CNT_EINVAL;
    return host_encode(n, n_len, out, out_words, flags, lut_from_of(n_len, flags, 27), 27, kChunkNt5, encode2_impl, invalid);
}
int cnt_bits_to_n2(const uint64_t* bits, size_t words, size_t len, uint8_t* out) {
    return host_decode(bits, words, len, out, 27, kChunkNt5, decode2_dev);
}

// ---- sharded tier -------------------------------------------------------------------
// Partition: shard_range() above.  No collective: outputs are disjoint ranges of `out`.
int cnt_n_to_bits_sharded_ex(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev, unsigned flags) {
    return sharded_host_encode(n, n_len, out, out_words, ndev, flags, 32, kChunkNt, encode_impl);
}
int cnt_n_to_bits_sharded(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev) {
    return cnt_n_to_bits_sharded_ex(n, n_len, out, out_words, ndev, 0);
}
int cnt_bits_to_n_sharded(const uint64_t* bits, size_t words, size_t len, uint8_t* out, int ndev) {
    return sharded_host_decode(bits, words, len, out, ndev, 32, cnt_bits_to_n);
}
// 5-letter codec over N GPUs: same scheme, shards are whole numbers of 128-word tiles (3456 nt)
int cnt_n_to_bits2_sharded_ex(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev, unsigned flags) {
    return sharded_host_encode(n, n_len, out, out_words, ndev, flags, 27, kChunkNt5, encode2_impl);
}
int cnt_n_to_bits2_sharded(const uint8_t* n, size_t n_len, uint64_t* out, size_t out_words, int ndev) {
    return cnt_n_to_bits2_sharded_ex(n, n_len, out, out_words, ndev, 0);
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
                              unsigned flags, float* shard_ms, int (*fn)(const void*, size_t, void*, size_t, unsigned, hipStream_t)) {
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

// the same work, enqueue-only: a caller-owned queue (one stream per shard), any number of ops queued ahead, one wait
int cnt_sharded_dev_open(int ndev, unsigned flags, void** queue) {
    ShardQueue* q = nullptr;
    if (!queue) return CNT_EINVAL;
    const int rc = shard_queue_open(ndev, flags, &q);
    *queue = q;
    return rc;
}
int cnt_sharded_dev_open_on_streams(int ndev, void* const* streams, unsigned flags, void** queue) {
    ShardQueue* q = nullptr;
    if (!queue) return CNT_EINVAL;
    const int rc = shard_queue_open(ndev, flags, &q, streams, true);
    *queue = q;
    return rc;
}
int cnt_sharded_dev_open_on_devices(int ndev, const int* devices, unsigned flags, void** queue) {
    ShardQueue* q = nullptr;
    if (!queue) return CNT_EINVAL;
    const int rc = shard_queue_open(ndev, flags, &q, nullptr, false, devices, true);
    *queue = q;
    return rc;
}
int cnt_sharded_dev_device(void* queue, int k, int* device) {
    ShardQueue* q = shard_queue_of(queue);
    if (!q || !device || k < 0 || k >= q->ndev) return CNT_EINVAL;
    *device = q->device_of(k);
    return CNT_OK;
}
int cnt_sharded_dev_wait_event(void* queue, int k, void* event) { return shard_queue_event(queue, k, event, false); }
int cnt_sharded_dev_record_event(void* queue, int k, void* event) { return shard_queue_event(queue, k, event, true); }
int cnt_sharded_dev_close(void* queue) { return shard_queue_close(queue); }
int cnt_sharded_dev_wait(void* queue, float* shard_ms) { return shard_queue_wait(queue, shard_ms); }
int cnt_sharded_dev_shards(void* queue, int* ndev) {
    ShardQueue* q = shard_queue_of(queue);
    if (!q || !ndev) return CNT_EINVAL;
    *ndev = q->ndev;
    return CNT_OK;
}
int cnt_sharded_dev_op_ms(void* queue, size_t op, float* shard_ms) {
    ShardQueue* q = shard_queue_of(queue);
    if (!q || !shard_ms) return CNT_EINVAL;
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    const int rc = q->op_times(op, shard_ms);
    (void)hipSetDevice(prev);
    return rc;
}
static int queue_encode(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags,
                        int (*fn)(const void*, size_t, void*, size_t, unsigned, hipStream_t)) {
    if (!d_n || !n_len || !d_out || !out_words) return CNT_EINVAL;
    return shard_queue_enqueue(queue, [&](int k, hipStream_t s) { return fn(d_n[k], n_len[k], d_out[k], out_words[k], flags, s); });
}
static int queue_decode(void* queue, const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, unsigned flags, dec_fn fn) {
    if (!d_bits || !words || !len || !d_out) return CNT_EINVAL;
    return shard_queue_enqueue(queue, [&](int k, hipStream_t s) { return fn(d_bits[k], words[k], len[k], d_out[k], flags, s); });
}
int cnt_n_to_bits_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags) {
    return queue_encode(queue, d_n, n_len, d_out, out_words, flags, encode_dev);
}
int cnt_bits_to_n_sharded_dev_enqueue(void* queue, const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, unsigned flags) {
    return queue_decode(queue, d_bits, words, len, d_out, flags, decode_dev);
}
int cnt_round_trip_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_bits, const size_t* out_words, void* const* d_back,
                                       unsigned flags) {
    if (!d_n || !n_len || !d_bits || !out_words || !d_back) return CNT_EINVAL;
    return shard_queue_enqueue(queue, [&](int k, hipStream_t s) { return round_trip_dev(d_n[k], n_len[k], d_bits[k], out_words[k], d_back[k], flags, s); });
}
int cnt_n_to_bits2_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags) {
    return queue_encode(queue, d_n, n_len, d_out, out_words, flags, encode2_dev);
}
// ... and the checked forms: d_invalid_count[k] is a device u64 ON SHARD k's DEVICE that the caller zeroes; shard k's op adds to it
static int queue_encode_checked(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags,
                                void* const* d_invalid, int (*fn)(const void*, size_t, void*, size_t, unsigned, void*, hipStream_t)) {
    if (!d_n || !n_len || !d_out || !out_words || !d_invalid) return CNT_EINVAL;
    return shard_queue_enqueue(queue, [&](int k, hipStream_t s) { return fn(d_n[k], n_len[k], d_out[k], out_words[k], flags, d_invalid[k], s); });
}
int cnt_n_to_bits_checked_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags,
                                              void* const* d_invalid_count) {
    return queue_encode_checked(queue, d_n, n_len, d_out, out_words, flags, d_invalid_count, encode_checked_dev);
}
int cnt_n_to_bits2_checked_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_out, const size_t* out_words, unsigned flags,
                                               void* const* d_invalid_count) {
    return queue_encode_checked(queue, d_n, n_len, d_out, out_words, flags, d_invalid_count, encode2_checked_dev);
}
int cnt_round_trip_checked_sharded_dev_enqueue(void* queue, const void* const* d_n, const size_t* n_len, void* const* d_bits, const size_t* out_words,
                                               void* const* d_back, unsigned flags, void* const* d_invalid_count) {
    if (!d_n || !n_len || !d_bits || !out_words || !d_back || !d_invalid_count) return CNT_EINVAL;
    return shard_queue_enqueue(queue, [&](int k, hipStream_t s) {
        return round_trip_checked_dev(d_n[k], n_len[k], d_bits[k], out_words[k], d_back[k], flags, d_invalid_count[k], s);
    });
}
int cnt_bits_to_n2_sharded_dev_enqueue(void* queue, const void* const* d_bits, const size_t* words, const size_t* len, void* const* d_out, unsigned flags) {
    return queue_decode(queue, d_bits, words, len, d_out, flags, decode2_dev);
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
// encode + validity count in ONE pass over the ASCII (1.25 B/nt instead of the 2.25 of cnt_validate_dev + cnt_n_to_bits_dev)
int cnt_n_to_bits_checked_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, void* d_invalid_count, void* stream) {
    return encode_checked_dev(d_n, n_len, d_out, out_words, flags, d_invalid_count, static_cast<hipStream_t>(stream));
}
int cnt_n_to_bits2_checked_dev(const void* d_n, size_t n_len, void* d_out, size_t out_words, unsigned flags, void* d_invalid_count, void* stream) {
    return encode2_checked_dev(d_n, n_len, d_out, out_words, flags, d_invalid_count, static_cast<hipStream_t>(stream));
}
int cnt_round_trip_checked_dev(const void* d_n, size_t n_len, void* d_bits, size_t out_words, void* d_back, unsigned flags, void* d_invalid_count, void* stream) {
    return round_trip_checked_dev(d_n, n_len, d_bits, out_words, d_back, flags, d_invalid_count, static_cast<hipStream_t>(stream));
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
#ifdef CNT_LAB_VARIANTS
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
    } else if (!strcmp(key, "round_trip_window_map")) {
        if (value < 0 || value > 2) return CNT_EINVAL;
        g_round_trip_window_map.store(value);
    } else if (!strcmp(key, "round_trip_plan")) {
        if (value < 0 || value > 3) return CNT_EINVAL;
        g_round_trip_plan.store(value);
    } else if (!strcmp(key, "decode_window")) {
        if (value < 0 || value > 1) return CNT_EINVAL;
        g_decode_window.store(value);
    } else if (!strcmp(key, "decode_cache_log2")) {
        if (value < -1 || value > 40) return CNT_EINVAL;
        g_decode_cache_log2.store(value);
    } else if (!strcmp(key, "decode_rot")) {
        if (value < -1 || (value > 3 && value != 10 && value != 11)) return CNT_EINVAL;
        g_decode_rot.store(value);
    } else if (!strcmp(key, "reduce_persistent")) {
        if (value < 0 || value > 1) return CNT_EINVAL;
        g_reduce_persistent.store(value);
    } else if (!strcmp(key, "hamming_order")) {
        if (value < 0 || value > 2) return CNT_EINVAL;
        g_hamming_order.store(value);
    } else if (!strcmp(key, "reduce_xi")) {
        if (value < 0 || value > 1) return CNT_EINVAL;
        g_reduce_xi.store(value);
    } else if (!strcmp(key, "xcd_shift")) {
        if (value < -1 || value > 6) return CNT_EINVAL;
        xcd_shift_override().store(value);
    } else if (!strcmp(key, "small_nt")) {
        if (value < 0) return CNT_EINVAL;
        g_small_nt.store(value);
    } else if (!strcmp(key, "launch_tiles")) {
        if (value < 0 || value % 64) return CNT_EINVAL;  // whole XCD-group permutations per launch
        launch_tiles_override().store(value);
    } else {
        return CNT_EINVAL;
    }
    return CNT_OK;
#else
    (void)value;
    return CNT_EINVAL;  // the product library has nothing to select: build bench/libcute_nt_hip_lab.so for A/B runs
#endif
}

int cnt_get_tuning(const char* key, int* value) {
    if (!key || !value) return CNT_EINVAL;
    if (!strcmp(key, "lab_build")) {
#ifdef CNT_LAB_VARIANTS
        *value = 1;
#else
        *value = 0;
#endif
    } else if (!strcmp(key, "encode")) *value = tune_encode();
    else if (!strcmp(key, "decode")) *value = tune_decode();
    else if (!strcmp(key, "encode2")) *value = tune_encode2();
    else if (!strcmp(key, "decode2")) *value = tune_decode2();
    else if (!strcmp(key, "small_nt")) *value = (int)tune_small_nt();
    else if (!strcmp(key, "xcd_shift")) *value = (int)xcd_shift();
    else if (!strcmp(key, "round_trip_cap")) *value = (int)tune_round_trip_cap();
    else if (!strcmp(key, "round_trip_shape")) *value = tune_round_trip_shape();
    else if (!strcmp(key, "encode_variants")) *value = kNumEncodeVariants;
    else if (!strcmp(key, "decode_variants")) *value = kNumDecodeVariants;
    else if (!strcmp(key, "encode2_variants")) *value = kNumEncode2Variants;
    else if (!strcmp(key, "decode2_variants")) *value = kNumDecode2Variants;
#ifdef CNT_LAB_VARIANTS
    else if (!strcmp(key, "launch_tiles")) *value = launch_tiles_override().load();
    else if (!strcmp(key, "decode_rot")) *value = g_decode_rot.load();
    else if (!strcmp(key, "decode_window")) *value = g_decode_window.load();
    else if (!strcmp(key, "decode_cache_log2")) *value = g_decode_cache_log2.load();
    else if (!strcmp(key, "reduce_xi")) *value = g_reduce_xi.load();
    else if (!strcmp(key, "hamming_order")) *value = g_hamming_order.load();
    else if (!strcmp(key, "reduce_fallbacks")) *value = g_reduce_fallbacks.load();
    else if (!strcmp(key, "reduce_persistent")) *value = g_reduce_persistent.load();
#else
    else if (!strcmp(key, "launch_tiles")) *value = 0;
    else if (!strcmp(key, "reduce_persistent")) *value = 1;
    else if (!strcmp(key, "reduce_fallbacks")) *value = 0;  // the form that could fall back is not in the product
#endif
    else return CNT_EINVAL;
    return CNT_OK;
}

int cnt_chip_info(int device, int* compute_units, int* lds_bytes_per_cu, int* xcds) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CNT_ENODEV;
    const ChipInfo c = query_chip(device);
    if (compute_units) *compute_units = (int)c.cus;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)c.lds_per_cu;
    if (xcds) *xcds = (int)c.xcds;
    return CNT_OK;
}

int cnt_chip_cache_nt(int device, uint64_t* nt) {
    int count = 0;
    if (!nt) return CNT_EINVAL;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return CNT_ENODEV;
    *nt = query_chip(device).cache_nt;
    return CNT_OK;
}

// Debug aid for FFI callers of the *_dev entry points (they take raw device pointers and only enqueue: a host pointer or a
// pointer of another device's memory is a GPU page fault -- a process abort -- when the kernel runs, not an error code).
int cnt_check_device_range(const void* p, size_t bytes, int device) {
    if (!p) return CNT_EINVAL;
    int count = 0, cur = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return CNT_ENODEV;
    if (device < 0) {
        HIP_TRY(hipGetDevice(&cur));
        device = cur;
    }
    if (device >= count) return CNT_ENODEV;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();
        return CNT_EINVAL;  // ordinary (pageable) host memory or not a mapping at all
    }
    if (a.type == hipMemoryTypeDevice && a.device != device) {
        int peer = 0;  // another device's memory: fine only where peer access is enabled -- the library never enables it
        if (hipDeviceCanAccessPeer(&peer, device, a.device) != hipSuccess || !peer) {
            (void)hipGetLastError();
            return CNT_EINVAL;
        }
    } else if (a.type != hipMemoryTypeDevice && a.type != hipMemoryTypeHost && a.type != hipMemoryTypeManaged) {
        return CNT_EINVAL;
    }
    if (bytes) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        const void* dp = a.type == hipMemoryTypeHost && a.devicePointer ? a.devicePointer : p;
        if (hipMemGetAddressRange(&base, &size, const_cast<void*>(dp)) != hipSuccess) {
            (void)hipGetLastError();
            return CNT_OK;  // pinned host memory registered by the caller has no range to ask for: the type check stands
        }
        const uintptr_t lo = reinterpret_cast<uintptr_t>(dp), end = reinterpret_cast<uintptr_t>(base) + size;
        if (lo + bytes < lo || lo + bytes > end) return CNT_ECAP;
    }
    return CNT_OK;
}

// ---- test hooks: compiled ONLY with -DCNT_TEST_HOOKS (tests/libcute_nt_hip_hooks.so and the lab build) -----------------------
// The product library exports none of them and holds no switch they could flip (include/cute_nt.h, "test support").
#ifdef CNT_TEST_HOOKS
int cnt_test_alias_devices(int on) { return g_alias_devices.exchange(on ? 1 : 0); }
int cnt_test_round_trip_plan(uint64_t a_n, uint64_t a_bits, uint64_t a_back, uint64_t n_len, unsigned flags, uint64_t* out) {
    if (!out || (a_bits & 7) || (flags & ~kEncodeFlags)) return CNT_EINVAL;
    const bool strict = (flags & CNT_STRICT_LUT) != 0;
    const uint64_t lut_from = lut_from_of(n_len, flags, 32);
    const uint64_t limit = lut_from != kNoLutWord && !strict ? n_len & ~(uint64_t)31 : n_len;
    const bool fast = !(a_n & 127) && !(a_bits & 127) && !(a_back & 127) && n_len >= kRoundTripTile;
    const RoundTripPlan p = fast ? RoundTripPlan{} : round_trip_plan((uintptr_t)a_n, (uintptr_t)a_bits, (uintptr_t)a_back, n_len, limit, tune_round_trip_plan());
    out[0] = fast ? 1 : 0;  // 1: the aligned kernel (round_trip_stream) takes the call
    out[1] = p.t0;
    out[2] = p.p0;
    out[3] = p.tiles;
    out[4] = (uint64_t)p.w0;
    out[5] = p.phase;
    out[6] = p.phase2;
    out[7] = kRoundTripAnySlackVecs;
    return CNT_OK;
}
int cnt_test_decode_plan(uint64_t a_bits, uint64_t a_out, uint64_t len, uint64_t cache_nt, uint64_t* out) {
    if (!out || (a_bits & 7) || !cache_nt) return CNT_EINVAL;
    const DecodePlan p = decode_plan((uintptr_t)a_bits, (uintptr_t)a_out, len, tune_decode_rot(), tune_decode_window(), cache_nt);
    out[0] = p.head;                                   // nucleotides in front of the first tile (edge items)
    out[1] = (a_out + p.head) & 4095;                  // the first tile's output byte inside its page (0 for len >= 2^20)
    out[2] = (a_bits + 4 * (p.head >> 4)) & 4095;      // the first tile's packed byte inside its page: where every XCD turn starts
    out[3] = p.sh;                                     // bit phase of the packed stream
    out[4] = p.q;                                      // dword phase against the 128-B line
    out[5] = p.window ? 1 : 0;                         // bits_to_n_window (else bits_to_n_stream)
    out[6] = p.tiles;
    return CNT_OK;
}
int cnt_test_pipeline_pieces(uint64_t total_nt, unsigned unit_nt, unsigned ramp_log2, uint64_t* out, int cap) {
    if ((unit_nt != 32 && unit_nt != 27) || !out || cap < 1) return -1;
    const size_t chunk = pipeline_chunk(total_nt, unit_nt, unit_nt == 32 ? kChunkNt : kChunkNt5);
    Pieces p(total_nt, chunk, (size_t)unit_nt * 8192, ramp_log2 ? (size_t)1 << ramp_log2 : 0);
    int n = 0;
    for (size_t m; (m = p.next()) != 0; ++n)
        if (n < cap) out[n] = m;
    return n;
}
int cnt_test_host_trace(int* tags, double* us, int cap) {
    const int n = (int)t_host_trace.size();
    for (int i = 0; i < n && i < cap; ++i) tags[i] = t_host_trace[(size_t)i].first, us[i] = t_host_trace[(size_t)i].second;
    return n;
}
int cnt_test_advise_output(void* out, size_t bytes) {
    if (!out) return CNT_EINVAL;
    advise_huge_output(out, bytes);
    return CNT_OK;
}
#endif  // CNT_TEST_HOOKS

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
