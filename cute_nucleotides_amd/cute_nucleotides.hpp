// cute_nucleotides.hpp -- C++ host-side mirror of the reference crate's public API over the
// C ABI (include/cute_nt.h).  Header-only; link with -lcute_nt_hip.
//
// The reference exposes (src/lib.rs:1-2)
//     cute_nucleotides::n_to_bits::{n_to_bits_*, bits_to_n_*}      src/n_to_bits.rs
//     cute_nucleotides::n_to_bits2::{n_to_bits2_*, bits_to_n2_*}   src/n_to_bits2.rs
// as `fn(&[u8]) -> Vec<u64>` / `fn(&[u64], usize) -> Vec<u8>`.  The same names with the
// `_hip` suffix live here in the same module layout, take spans (pointer + length) and return
// owned vectors, exactly like the Rust functions: callee allocates, caller only lends.
//
// Error behaviour: `len` larger than the packed capacity throws std::length_error with the
// reference's panic text (n_to_bits.rs:52-54); any other failure (HIP error, no device)
// throws std::runtime_error carrying cnt_strerror().  Like the reference's functions these
// are re-entrant and callable from any thread.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../include/cute_nt.h"

namespace cute_nucleotides {

namespace detail {
// The reference's functions allocate their result UNINITIALISED and set its length afterwards (Vec::with_capacity /
// alloc + set_len / from_raw_parts, n_to_bits.rs:88-89,113): the output is written exactly once, by the codec.
// std::vector<T>(n) would first zero-fill it -- single-threaded, page by page: a 1-GiB decode through this mirror cost
// 262 ms against 85 ms for the C call into a fresh malloc (profiles/r03_bench_twin.log).  This allocator makes
// `Vec<T>(n)` default-initialise (= leave alone) trivially constructible elements; everything else is std::allocator.
template <class T>
struct uninit_allocator : std::allocator<T> {
    template <class U>
    struct rebind {
        using other = uninit_allocator<U>;
    };
    uninit_allocator() = default;
    template <class U>
    uninit_allocator(const uninit_allocator<U>&) noexcept {}
    template <class U>
    void construct(U* p) noexcept(noexcept(::new (static_cast<void*>(p)) U)) {
        ::new (static_cast<void*>(p)) U;  // default-initialisation: no zero fill for u8 / u64
    }
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        ::new (static_cast<void*>(p)) U(std::forward<A>(a)...);
    }
};
// Vec<T> against a plain std::vector<T> (found by ADL through the allocator's namespace)
template <class T, class B, std::enable_if_t<!std::is_same<B, uninit_allocator<T>>::value, int> = 0>
inline bool operator==(const std::vector<T, uninit_allocator<T>>& a, const std::vector<T, B>& b) {
    return a.size() == b.size() && std::equal(a.begin(), a.end(), b.begin());
}
template <class T, class B, std::enable_if_t<!std::is_same<B, uninit_allocator<T>>::value, int> = 0>
inline bool operator!=(const std::vector<T, uninit_allocator<T>>& a, const std::vector<T, B>& b) { return !(a == b); }
template <class T>
inline bool operator==(const std::vector<T>& a, const std::vector<T, uninit_allocator<T>>& b) { return b == a; }
template <class T>
inline bool operator!=(const std::vector<T>& a, const std::vector<T, uninit_allocator<T>>& b) { return !(b == a); }
inline void check(int status) {
    if (status == CNT_OK) return;
    if (status == CNT_ELEN) throw std::length_error(cnt_strerror(status));  // the reference's panic
    throw std::runtime_error(std::string("libcute_nt_hip: ") + cnt_strerror(status));
}
}  // namespace detail

/// What the `*_hip` functions return: a std::vector whose elements are NOT zero-filled on construction (the
/// reference's `Vec` with `set_len`).  Converts to a plain std::vector with `std::vector<T>(v.begin(), v.end())`.
template <class T>
using Vec = std::vector<T, detail::uninit_allocator<T>>;

namespace n_to_bits {

/// Encode {A,T/U,C,G} -> {00,10,01,11}, 32 nt per u64, LSB first (n_to_bits.rs:34-47 and the
/// four SIMD siblings).  `strict_lut` selects n_to_bits_lut's table semantics for bytes outside
/// the alphabet (they encode as 0) instead of the SIMD variants' (byte>>1)&3; `tail_lut` is the SIMD variants to
/// the letter: bit extraction on whole 32-nt blocks, the table on the final partial word (n_to_bits.rs:109-111).
inline Vec<uint64_t> n_to_bits_hip(const uint8_t* n, size_t len, bool strict_lut = false, bool tail_lut = false) {
    Vec<uint64_t> out(cnt_words_for(len));
    detail::check(cnt_n_to_bits_ex(n, len, out.data(), out.size(), (strict_lut ? CNT_STRICT_LUT : 0u) | (tail_lut ? CNT_TAIL_LUT : 0u)));
    return out;
}
template <class A>
inline Vec<uint64_t> n_to_bits_hip(const std::vector<uint8_t, A>& n) { return n_to_bits_hip(n.data(), n.size()); }

/// The same encode VALIDATED in the same pass over the data: `invalid` receives the number of bytes outside ACGTUacgtu -- what
/// the reference's BYTE_LUT turns into code 0 without a word (n_to_bits.rs:8-21,42; README.md:23 points at a separate check).
/// The words are n_to_bits_hip's whatever the count says.
inline Vec<uint64_t> n_to_bits_hip_checked(const uint8_t* n, size_t len, uint64_t& invalid, bool strict_lut = false, bool tail_lut = false) {
    Vec<uint64_t> out(cnt_words_for(len));
    detail::check(cnt_n_to_bits_checked(n, len, out.data(), out.size(), (strict_lut ? CNT_STRICT_LUT : 0u) | (tail_lut ? CNT_TAIL_LUT : 0u), &invalid));
    return out;
}

/// Decode `len` nucleotides (n_to_bits.rs:51-69 and the three SIMD siblings).
inline Vec<uint8_t> bits_to_n_hip(const uint64_t* bits, size_t words, size_t len) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    Vec<uint8_t> out(len);
    detail::check(cnt_bits_to_n(bits, words, len, out.data()));
    return out;
}
template <class A>
inline Vec<uint8_t> bits_to_n_hip(const std::vector<uint64_t, A>& bits, size_t len) {
    return bits_to_n_hip(bits.data(), bits.size(), len);
}

/// `_into` forms: the same codecs writing into a vector the CALLER owns (resized without zero-fill, capacity reused).  A
/// loop that times like the reference's harness -- result allocated AND dropped inside the timed call,
/// benches/bench_n_to_bits.rs:6-7 -- then pays neither the page faults of a fresh allocation nor the munmap of the
/// dropped one (1-GiB decode: 22 ms instead of 74 ms, BENCH_r03 host_tier).  Same results and exceptions.
inline void n_to_bits_hip_into(const uint8_t* n, size_t len, Vec<uint64_t>& out, bool strict_lut = false, bool tail_lut = false) {
    out.resize(cnt_words_for(len));
    detail::check(cnt_n_to_bits_ex(n, len, out.data(), out.size(), (strict_lut ? CNT_STRICT_LUT : 0u) | (tail_lut ? CNT_TAIL_LUT : 0u)));
}
inline void bits_to_n_hip_into(const uint64_t* bits, size_t words, size_t len, Vec<uint8_t>& out) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    out.resize(len);
    detail::check(cnt_bits_to_n(bits, words, len, out.data()));
}

/// Span forms: the 2-bit codec between buffers the CALLER owns -- the forms that can use PINNED memory (`PinnedBuffer`,
/// `HostPin`): a side that lies in pinned memory is not staged, the copy engines read / write it in place (include/cute_nt.h
/// "pinned caller memory").  `out` holds at least cnt_words_for(len) words / `len` bytes; returns the count written.
inline size_t n_to_bits_hip_into(const uint8_t* n, size_t len, uint64_t* out, size_t out_words, bool strict_lut = false, bool tail_lut = false) {
    detail::check(cnt_n_to_bits_ex(n, len, out, out_words, (strict_lut ? CNT_STRICT_LUT : 0u) | (tail_lut ? CNT_TAIL_LUT : 0u)));
    return cnt_words_for(len);
}
inline size_t bits_to_n_hip_into(const uint64_t* bits, size_t words, size_t len, uint8_t* out) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    detail::check(cnt_bits_to_n(bits, words, len, out));
    return len;
}

/// `count` elements of pinned, device-mapped host memory (cnt_host_alloc; NOT zeroed).  For buffers that live as long as the
/// pipeline: pinned memory cannot be swapped.
template <class T>
class PinnedBuffer {
   public:
    explicit PinnedBuffer(size_t count) : count_(count) {
        void* p = nullptr;
        detail::check(cnt_host_alloc(&p, count * sizeof(T)));
        ptr_ = static_cast<T*>(p);
    }
    ~PinnedBuffer() { (void)cnt_host_free(ptr_); }
    PinnedBuffer(const PinnedBuffer&) = delete;
    PinnedBuffer& operator=(const PinnedBuffer&) = delete;
    PinnedBuffer(PinnedBuffer&& o) noexcept : ptr_(o.ptr_), count_(o.count_) { o.ptr_ = nullptr, o.count_ = 0; }
    T* data() { return ptr_; }
    const T* data() const { return ptr_; }
    size_t size() const { return count_; }
    T& operator[](size_t i) { return ptr_[i]; }
    const T& operator[](size_t i) const { return ptr_[i]; }
    T* begin() { return ptr_; }
    T* end() { return ptr_ + count_; }

   private:
    T* ptr_ = nullptr;
    size_t count_ = 0;
};

/// Pins an EXISTING range in place for the guard's lifetime (cnt_host_register: tens of milliseconds per GiB, once).
class HostPin {
   public:
    HostPin(void* p, size_t bytes) : p_(p) { detail::check(cnt_host_register(p, bytes)); }
    ~HostPin() { (void)cnt_host_unregister(p_); }
    HostPin(const HostPin&) = delete;
    HostPin& operator=(const HostPin&) = delete;

   private:
    void* p_;
};

/// would the host tier use [p, p + bytes) in place?
inline bool is_pinned(const void* p, size_t bytes) { return cnt_host_is_pinned(p, bytes) == 1; }

/// The same, cut into contiguous chunks over `ndev` GPUs (<= 0: all visible), no collective.
inline Vec<uint64_t> n_to_bits_hip_sharded(const uint8_t* n, size_t len, int ndev = 0) {
    Vec<uint64_t> out(cnt_words_for(len));
    detail::check(cnt_n_to_bits_sharded(n, len, out.data(), out.size(), ndev));
    return out;
}
inline Vec<uint8_t> bits_to_n_hip_sharded(const uint64_t* bits, size_t words, size_t len, int ndev = 0) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    Vec<uint8_t> out(len);
    detail::check(cnt_bits_to_n_sharded(bits, words, len, out.data(), ndev));
    return out;
}

}  // namespace n_to_bits

namespace n_to_bits2 {

/// 5-letter codec, 3 nt -> 7 bits, 27 nt per u64 (n_to_bits2.rs:37-74, :118-189).
inline Vec<uint64_t> n_to_bits2_hip(const uint8_t* n, size_t len) {
    Vec<uint64_t> out(cnt_words2_for(len));
    detail::check(cnt_n_to_bits2(n, len, out.data(), out.size()));
    return out;
}
template <class A>
inline Vec<uint64_t> n_to_bits2_hip(const std::vector<uint8_t, A>& n) { return n_to_bits2_hip(n.data(), n.size()); }

/// ... validated in the same pass: `invalid` = bytes outside ACGTUNacgtun (n_to_bits2.rs:8-23)
inline Vec<uint64_t> n_to_bits2_hip_checked(const uint8_t* n, size_t len, uint64_t& invalid) {
    Vec<uint64_t> out(cnt_words2_for(len));
    detail::check(cnt_n_to_bits2_checked(n, len, out.data(), out.size(), 0u, &invalid));
    return out;
}

/// n_to_bits2.rs:78-107, :196-268.
inline Vec<uint8_t> bits_to_n2_hip(const uint64_t* bits, size_t words, size_t len) {
    if (words > SIZE_MAX / 27 || len > words * 27) detail::check(CNT_ELEN);
    Vec<uint8_t> out(len);
    detail::check(cnt_bits_to_n2(bits, words, len, out.data()));
    return out;
}
template <class A>
inline Vec<uint8_t> bits_to_n2_hip(const std::vector<uint64_t, A>& bits, size_t len) {
    return bits_to_n2_hip(bits.data(), bits.size(), len);
}

inline void n_to_bits2_hip_into(const uint8_t* n, size_t len, Vec<uint64_t>& out) {
    out.resize(cnt_words2_for(len));
    detail::check(cnt_n_to_bits2(n, len, out.data(), out.size()));
}
inline void bits_to_n2_hip_into(const uint64_t* bits, size_t words, size_t len, Vec<uint8_t>& out) {
    if (words > SIZE_MAX / 27 || len > words * 27) detail::check(CNT_ELEN);
    out.resize(len);
    detail::check(cnt_bits_to_n2(bits, words, len, out.data()));
}

/// The 5-letter codec over `ndev` GPUs (shards are whole 128-word tiles), no collective.
inline Vec<uint64_t> n_to_bits2_hip_sharded(const uint8_t* n, size_t len, int ndev = 0) {
    Vec<uint64_t> out(cnt_words2_for(len));
    detail::check(cnt_n_to_bits2_sharded(n, len, out.data(), out.size(), ndev));
    return out;
}
inline Vec<uint8_t> bits_to_n2_hip_sharded(const uint64_t* bits, size_t words, size_t len, int ndev = 0) {
    if (words > SIZE_MAX / 27 || len > words * 27) detail::check(CNT_ELEN);
    Vec<uint8_t> out(len);
    detail::check(cnt_bits_to_n2_sharded(bits, words, len, out.data(), ndev));
    return out;
}

}  // namespace n_to_bits2

/// Device-resident tier for C++ callers that do not link HIP themselves (the same shape as the Rust binding's
/// `DeviceBuffer` / `n_to_bits_hip_dev`, rust/src/hip.rs): data already in HBM, enqueue on the default stream,
/// `sync()` waits.  This is the tier the roofline numbers measure.
namespace device {

class DeviceBuffer {
   public:
    explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { detail::check(cnt_dev_alloc(&ptr_, bytes)); }
    DeviceBuffer(const void* host, size_t bytes) : DeviceBuffer(bytes) { detail::check(cnt_dev_upload(ptr_, host, bytes)); }
    ~DeviceBuffer() { (void)cnt_dev_free(ptr_); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    void* data() const { return ptr_; }
    size_t size_bytes() const { return bytes_; }
    template <typename T>
    Vec<T> to_vector(size_t count) const {  // synchronous copy back to the host
        if (count * sizeof(T) > bytes_) throw std::out_of_range("DeviceBuffer::to_vector");
        Vec<T> out(count);
        detail::check(cnt_dev_download(out.data(), ptr_, count * sizeof(T)));
        return out;
    }

   private:
    void* ptr_ = nullptr;
    size_t bytes_ = 0;
};

/// Enqueue the encode of `n_len` resident nucleotides into `out` (>= cnt_words_for(n_len) words).
inline void n_to_bits_hip_dev(const DeviceBuffer& n, size_t n_len, DeviceBuffer& out, bool strict_lut = false) {
    if (n_len > n.size_bytes()) throw std::out_of_range("n_to_bits_hip_dev: n_len");
    detail::check(cnt_n_to_bits_dev(n.data(), n_len, out.data(), out.size_bytes() / 8, strict_lut ? CNT_STRICT_LUT : 0u, nullptr));
}
/// Encode + validity count in ONE pass over the resident ASCII (cnt_n_to_bits_checked_dev): the launch ADDS the number of
/// bytes outside ACGTUacgtu to the u64 at `invalid` (device memory the caller zeroed, e.g. an 8-byte DeviceBuffer).
/// `spread` (CNT_SPREAD_COUNT): `invalid` holds CNT_COUNT_SLOTS u64 (16 KiB) and the count is their sum -- for dirty data.
inline void n_to_bits_hip_checked_dev(const DeviceBuffer& n, size_t n_len, DeviceBuffer& out, DeviceBuffer& invalid, bool strict_lut = false, bool spread = false) {
    if (n_len > n.size_bytes() || invalid.size_bytes() < (spread ? 8u * CNT_COUNT_SLOTS : 8u)) throw std::out_of_range("n_to_bits_hip_checked_dev: sizes");
    detail::check(cnt_n_to_bits_checked_dev(n.data(), n_len, out.data(), out.size_bytes() / 8, (strict_lut ? CNT_STRICT_LUT : 0u) | (spread ? CNT_SPREAD_COUNT : 0u),
                                            invalid.data(), nullptr));
}
/// Enqueue the decode of `len` nucleotides from `words` resident words into `out` (>= len bytes).
inline void bits_to_n_hip_dev(const DeviceBuffer& bits, size_t words, size_t len, DeviceBuffer& out) {
    if (len > (words << 5)) detail::check(CNT_ELEN);
    if (words * 8 > bits.size_bytes() || len > out.size_bytes()) throw std::out_of_range("bits_to_n_hip_dev: sizes");
    detail::check(cnt_bits_to_n_dev(bits.data(), words, len, out.data(), 0u, nullptr));
}
inline void sync() { detail::check(cnt_dev_sync(nullptr)); }
inline void set_device(int device) { detail::check(cnt_set_device(device)); }  // what DeviceBuffer allocates on

/// Enqueue-only multi-GPU device tier (cnt_sharded_dev_open / *_enqueue / cnt_sharded_dev_wait): shard k resident on
/// device k, one library stream per shard; every enqueue_* call queues that codec call on every shard and returns at
/// once, wait() drains all of them.  Queue the decode of step s behind its encode, step s+1 behind that, wait once: the
/// devices run back to back with the host a whole queue ahead (word w depends on nucleotides [32w, 32w+32) only,
/// n_to_bits.rs:38-43 -- nothing needs a barrier).
class ShardedDevQueue {
   public:
    explicit ShardedDevQueue(int ndev, bool timed = false) {
        detail::check(cnt_sharded_dev_open(ndev, timed ? CNT_QUEUE_TIMED : 0u, &handle_));
        detail::check(cnt_sharded_dev_shards(handle_, &ndev_));
    }
    /// the queue ADOPTS the caller's streams (streams[k]: a non-null hipStream_t of device k): shard k's ops run in order with
    /// whatever else the caller enqueues there; close() leaves them alone
    explicit ShardedDevQueue(const std::vector<void*>& streams, bool timed = false) {
        detail::check(cnt_sharded_dev_open_on_streams((int)streams.size(), streams.data(), timed ? CNT_QUEUE_TIMED : 0u, &handle_));
        detail::check(cnt_sharded_dev_shards(handle_, &ndev_));
    }
    /// library-owned streams on an explicit device list: shard k runs on devices[k] (any subset, order or repetition)
    static ShardedDevQueue on_devices(const std::vector<int>& devices, bool timed = false) {
        void* h = nullptr;
        detail::check(cnt_sharded_dev_open_on_devices((int)devices.size(), devices.data(), timed ? CNT_QUEUE_TIMED : 0u, &h));
        return ShardedDevQueue(h);
    }
    ShardedDevQueue(ShardedDevQueue&& o) noexcept : handle_(o.handle_), ndev_(o.ndev_) { o.handle_ = nullptr; }
    ~ShardedDevQueue() {
        if (handle_) (void)cnt_sharded_dev_close(handle_);
    }
    /// the device the queue runs shard k on -- for adopted streams what the STREAM says (hipStreamGetDevice), not k
    int device(int k) const {
        int d = -1;
        detail::check(cnt_sharded_dev_device(handle_, k, &d));
        return d;
    }
    /// the validated encode on every shard: shard k's op adds its count of bytes outside ACGTUacgtu to the u64 in invalid[k]
    /// (8 bytes on shard k's device, zeroed by the caller)
    void enqueue_n_to_bits_checked(const std::vector<const DeviceBuffer*>& n, const std::vector<size_t>& n_len, const std::vector<DeviceBuffer*>& out,
                                   const std::vector<DeviceBuffer*>& invalid, bool strict_lut = false) {
        if ((int)n.size() != ndev_ || (int)n_len.size() != ndev_ || (int)out.size() != ndev_ || (int)invalid.size() != ndev_) throw std::invalid_argument("one entry per shard");
        std::vector<const void*> in(n.size());
        std::vector<void*> o(n.size()), c(n.size());
        std::vector<size_t> cap(n.size());
        for (size_t k = 0; k < n.size(); ++k) {
            if (n_len[k] > n[k]->size_bytes() || invalid[k]->size_bytes() < 8) throw std::out_of_range("enqueue_n_to_bits_checked: sizes");
            in[k] = n[k]->data();
            o[k] = out[k]->data();
            c[k] = invalid[k]->data();
            cap[k] = out[k]->size_bytes() / 8;
        }
        detail::check(cnt_n_to_bits_checked_sharded_dev_enqueue(handle_, in.data(), n_len.data(), o.data(), cap.data(), strict_lut ? CNT_STRICT_LUT : 0u, c.data()));
    }
    /// shard k's stream waits ON THE DEVICE for `event` (a hipEvent_t recorded behind the producer of shard k's buffer)
    void wait_event(int k, void* event) { detail::check(cnt_sharded_dev_wait_event(handle_, k, event)); }
    /// records the caller's hipEvent_t (of shard k's device) behind everything queued for shard k so far
    void record_event(int k, void* event) { detail::check(cnt_sharded_dev_record_event(handle_, k, event)); }
    /// the fused encode + decode of cnt_round_trip_dev on every shard
    void enqueue_round_trip(const std::vector<const DeviceBuffer*>& n, const std::vector<size_t>& n_len, const std::vector<DeviceBuffer*>& bits,
                            const std::vector<DeviceBuffer*>& back, bool strict_lut = false) {
        if ((int)n.size() != ndev_ || (int)n_len.size() != ndev_ || (int)bits.size() != ndev_ || (int)back.size() != ndev_) throw std::invalid_argument("one entry per shard");
        std::vector<const void*> in(n.size());
        std::vector<void*> o(n.size()), b(n.size());
        std::vector<size_t> cap(n.size());
        for (size_t k = 0; k < n.size(); ++k) {
            if (n_len[k] > n[k]->size_bytes() || n_len[k] > back[k]->size_bytes()) throw std::out_of_range("enqueue_round_trip: n_len");
            in[k] = n[k]->data();
            o[k] = bits[k]->data();
            b[k] = back[k]->data();
            cap[k] = bits[k]->size_bytes() / 8;
        }
        detail::check(cnt_round_trip_sharded_dev_enqueue(handle_, in.data(), n_len.data(), o.data(), cap.data(), b.data(), strict_lut ? CNT_STRICT_LUT : 0u));
    }
    ShardedDevQueue(const ShardedDevQueue&) = delete;
    ShardedDevQueue& operator=(const ShardedDevQueue&) = delete;
    int shards() const { return ndev_; }
    void enqueue_n_to_bits(const std::vector<const DeviceBuffer*>& n, const std::vector<size_t>& n_len, const std::vector<DeviceBuffer*>& out, bool strict_lut = false) {
        if ((int)n.size() != ndev_ || (int)n_len.size() != ndev_ || (int)out.size() != ndev_) throw std::invalid_argument("one entry per shard");
        std::vector<const void*> in(n.size());
        std::vector<void*> o(n.size());
        std::vector<size_t> cap(n.size());
        for (size_t k = 0; k < n.size(); ++k) {
            if (n_len[k] > n[k]->size_bytes()) throw std::out_of_range("enqueue_n_to_bits: n_len");
            in[k] = n[k]->data();
            o[k] = out[k]->data();
            cap[k] = out[k]->size_bytes() / 8;
        }
        detail::check(cnt_n_to_bits_sharded_dev_enqueue(handle_, in.data(), n_len.data(), o.data(), cap.data(), strict_lut ? CNT_STRICT_LUT : 0u));
    }
    void enqueue_bits_to_n(const std::vector<const DeviceBuffer*>& bits, const std::vector<size_t>& words, const std::vector<size_t>& len, const std::vector<DeviceBuffer*>& out) {
        if ((int)bits.size() != ndev_ || (int)words.size() != ndev_ || (int)len.size() != ndev_ || (int)out.size() != ndev_) throw std::invalid_argument("one entry per shard");
        std::vector<const void*> in(bits.size());
        std::vector<void*> o(bits.size());
        for (size_t k = 0; k < bits.size(); ++k) {
            if (len[k] > (words[k] << 5)) detail::check(CNT_ELEN);
            if (words[k] * 8 > bits[k]->size_bytes() || len[k] > out[k]->size_bytes()) throw std::out_of_range("enqueue_bits_to_n: sizes");
            in[k] = bits[k]->data();
            o[k] = out[k]->data();
        }
        detail::check(cnt_bits_to_n_sharded_dev_enqueue(handle_, in.data(), words.data(), len.data(), o.data(), 0u));
    }
    /// waits for everything queued; per-shard device milliseconds of the batch (zeros unless timed)
    std::vector<float> wait() {
        std::vector<float> ms((size_t)ndev_, 0.f);
        detail::check(cnt_sharded_dev_wait(handle_, ms.data()));
        return ms;
    }
    std::vector<float> op_ms(size_t op) const {
        std::vector<float> ms((size_t)ndev_, 0.f);
        detail::check(cnt_sharded_dev_op_ms(handle_, op, ms.data()));
        return ms;
    }

   private:
    explicit ShardedDevQueue(void* adopted_handle) : handle_(adopted_handle) { detail::check(cnt_sharded_dev_shards(handle_, &ndev_)); }
    void* handle_ = nullptr;
    int ndev_ = 0;
};

}  // namespace device

}  // namespace cute_nucleotides
