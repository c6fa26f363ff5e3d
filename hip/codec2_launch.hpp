// codec2_launch.hpp -- the kernel-variant tables of the 2-bit codec and their launchers.
//
// Variant 0 of each direction is the shipped default (the measured best on MI355X at the metric size, 2^34 nt) and the
// ONLY one the product library (libcute_nt_hip.so) contains.  Everything else -- the other shapes, cache policies, tile
// maps and residency caps that the A/B numbers in profiles/ were measured with, and the process-global knobs that select
// them -- is compiled only with -DCNT_LAB_VARIANTS, into bench/libcute_nt_hip_lab.so (cute_nucleotides_amd/build.py
// build_lab()): the product has no mutable kernel selection, like the reference's pure functions over immutable tables
// (n_to_bits.rs:8,23).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <thread>

#include "codec2_kernels.hpp"

namespace cnt {

struct VariantDesc {
    const char* name;
    uint32_t tile_nt;  // nucleotides per workgroup
    uint32_t block;    // threads per workgroup (bounds the tiles one launch may cover)
    uint32_t wg_cap;   // resident workgroups per CU to aim for via a dummy LDS allocation (0 = no cap)
};

// What the launch geometry depends on, asked of the device instead of assumed (an MI355X in SPX mode answers
// 256 CUs / 160 KiB of LDS per CU / 8 XCDs; the CPX / DPX / QPX partition modes, or another CDNA part, do not):
//   cus         multiProcessorCount                  -> grid of the persistent reductions
//   lds_per_cu  maxSharedMemoryPerMultiProcessor     -> the dummy-LDS residency caps
//   xcd_shift   log2(hipDeviceAttributeNumberOfXccs) -> the XCD-aware block -> tile maps (0 = identity map when the
//               count is unknown or not a power of two)
// Cached per device index; correctness never depends on any of it.
//   cache_nt    nucleotides whose PACKED form (a quarter of a byte each) fills this device's share of the memory-side Infinity
//               Cache: 32 MiB per XCD -> 2^27 nt per XCD, 2^30 nt on an SPX MI355X (8 XCDs, 256 MiB).  No HIP attribute reports
//               the cache; the XCD count does, and a partition of fewer XCDs competes for it with its siblings.  Decode's
//               launch plan treats calls beyond it as streaming from HBM (device_tier.inc decode_plan).
struct ChipInfo {
    uint32_t cus, lds_per_cu, xcds, xcd_shift;
    uint64_t cache_nt;
};
inline uint64_t decode_cache_nt_of(uint32_t xcds) { return (uint64_t)(xcds ? xcds : 1) << 27; }
inline ChipInfo query_chip(int device) {
    ChipInfo c{0, 0, 0, 0, 0};
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c.cus = (uint32_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && v > 0) c.lds_per_cu = (uint32_t)v;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeNumberOfXccs, device) == hipSuccess && v > 0) c.xcds = (uint32_t)v;
    (void)hipGetLastError();
    if (!c.cus) c.cus = 1;
    if (c.lds_per_cu < 65536u) c.lds_per_cu = 65536u;  // every CDNA CU has at least 64 KiB; a smaller answer is a per-block limit
    if (!c.xcds) c.xcds = 1;
    if ((c.xcds & (c.xcds - 1)) == 0)
        while ((1u << c.xcd_shift) < c.xcds) ++c.xcd_shift;
    c.cache_nt = decode_cache_nt_of(c.xcds);
    return c;
}
inline const ChipInfo& chip_info() {  // of the calling thread's current device
    constexpr int kMaxDev = 64;
    static ChipInfo table[kMaxDev];
    static std::atomic<uint8_t> state[kMaxDev];  // 0 = empty, 1 = one thread is filling it, 2 = ready
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev >= kMaxDev) {  // beyond the cache: ask every time rather than quote device 0's geometry
        static thread_local ChipInfo uncached;
        uncached = query_chip(dev);
        return uncached;
    }
    if (state[dev].load(std::memory_order_acquire) != 2) {
        uint8_t expect = 0;
        if (state[dev].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) {
            table[dev] = query_chip(dev);  // exactly one writer
            state[dev].store(2, std::memory_order_release);
        } else {
            while (state[dev].load(std::memory_order_acquire) != 2) std::this_thread::yield();  // three attribute queries away
        }
    }
    return table[dev];
}

// log2 of the XCD count the block -> tile maps are built for: the device's.  Lab build: the tuning key "xcd_shift" overrides it
// (A/B runs, and the test that walks every value a partition mode could produce: the maps must be bijections for ANY value).
#ifdef CNT_LAB_VARIANTS
inline std::atomic<int>& xcd_shift_override() {
    static std::atomic<int> v{-1};
    return v;
}
inline uint32_t xcd_shift() {
    const int o = xcd_shift_override().load(std::memory_order_relaxed);
    return o >= 0 ? (uint32_t)o : chip_info().xcd_shift;
}
#else
inline uint32_t xcd_shift() { return chip_info().xcd_shift; }
#endif

// dynamic-LDS bytes that let `cap` workgroups (and no more) fit in a CU's LDS (160 KiB on gfx950)
inline uint32_t lds_for_cap(uint32_t cap) { return cap ? (chip_info().lds_per_cu / cap) / 256u * 256u : 0u; }
// the same for a kernel that already owns `static_bytes` of LDS: the dynamic part that completes the cap, saturating (on a
// 64-KiB-LDS part a 10-per-CU cap is 6400 B, below the 7168-B slab of the four-words-per-lane variants: no padding then)
inline uint32_t lds_pad_for_cap(uint32_t cap, uint32_t static_bytes) {
    const uint32_t total = lds_for_cap(cap);
    return total > static_bytes ? total - static_bytes : 0u;
}

// ---- encode -------------------------------------------------------------------------
constexpr VariantDesc kEncodeVariants[] = {
    {"stream B=64 U=2 plain order ld=nt st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 0: default (round 3; rounds 1-2: variant 17)
#ifdef CNT_LAB_VARIANTS
    {"stream B=256 U=1 ld=nt st=sc1", 256 * 1 * 16, 256, 0},              // 1
    {"stream B=512 U=1 ld=sc0|nt st=sc1", 512 * 1 * 16, 512, 0},          // 2
    {"stream B=256 U=4 ld=nt st=nt", 256 * 4 * 16, 256, 0},               // 3: the first shape tried
    {"lds B=256 U=4 ld=nt st=sc1", 256 * 4 * 16, 256, 0},                 // 4: LDS-widened stores
    {"stream B=64 U=2 ld=nt st=sc1", 64 * 2 * 16, 64, 0},                // 5: plain order, write-through-only stores, no residency cap
    {"stream B=128 U=2 ld=nt st=sc1", 128 * 2 * 16, 128, 0},              // 6
    {"stream B=64 U=2 xcd-quads ld=nt st=sc1", 64 * 2 * 16, 64, 0},      // 7
    {"stream B=256 U=4 plain", 256 * 4 * 16, 256, 0},                     // 8: no cache-policy bits at all
    {"stream B=64 U=2 xcd-pairs ld=nt st=sc1 (no residency cap)", 64 * 2 * 16, 64, 0},  // 9: as 11, uncapped
    {"stream B=128 U=2 ld=nt st=sc1, 10 wg/CU", 128 * 2 * 16, 128, 10},   // 10
    {"stream B=64 U=2 xcd-pairs ld=nt st=sc1, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 11: as 0 with write-through-only stores
    {"stream B=64 U=2 xcd-pairs ld=sc0|nt st=sc1, 22 wg/CU", 64 * 2 * 16, 64, 22},  // 12
    {"stream B=64 U=2 xcd-pairs ld=sc0|nt st=sc0|sc1|nt, 22 wg/CU", 64 * 2 * 16, 64, 22},  // 13
    // round 3: device-scope loads (sc1: do not allocate in the CU's vector L1) -- what moved decode by ~1 % (its variant 20)
    {"stream B=64 U=2 xcd-pairs ld=sc1|nt st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 14
    {"stream B=64 U=2 xcd-pairs ld=sc1 st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},     // 15
    {"stream B=64 U=2 xcd-pairs ld=sc0|sc1|nt st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 16
    // round 3, from the xcd_shift A/B (bench/xcd_shift_ab.py): the policies and cap of rounds 1-2's default WITHOUT the XCD
    // pairing is a combination the ladder never measured (the pairing was adopted before the cap and the store policy
    // were) -- and in encode -> decode steps it is 0.2-0.9 % FASTER on five boxes out of five
    // (profiles/r03_ab_step_encode_plain_order.log): it became variant 0, the pairing moved here
    {"stream B=64 U=2 xcd-pairs ld=nt st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 17: the default of rounds 1-2
    {"stream B=64 U=2 plain order ld=nt st=sc0|sc1|nt, 22 wg/CU", 64 * 2 * 16, 64, 22},  // 18
    {"stream B=64 U=2 plain order ld=nt st=sc0|sc1|nt, 24 wg/CU", 64 * 2 * 16, 64, 24},  // 19
    {"stream B=64 U=2 plain order ld=nt st=sc0|sc1|nt, 26 wg/CU", 64 * 2 * 16, 64, 26},  // 20
    {"stream B=64 U=2 plain order ld=nt st=sc0|sc1|nt, 20 wg/CU", 64 * 2 * 16, 64, 20},  // 21
    // the policies once more, now around the plain-order default
    {"stream B=64 U=2 plain order ld=plain st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},   // 22
    {"stream B=64 U=2 plain order ld=sc1|nt st=sc0|sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},  // 23
    {"stream B=64 U=2 plain order ld=nt st=sc1|nt, 23 wg/CU", 64 * 2 * 16, 64, 23},          // 24
    {"stream B=64 U=2 plain order ld=nt st=sc1, 23 wg/CU", 64 * 2 * 16, 64, 23},             // 25
    {"stream B=128 U=2 plain order ld=nt st=sc0|sc1|nt, 11 wg/CU", 128 * 2 * 16, 128, 11},   // 26
    {"stream B=64 U=4 plain order ld=nt st=sc0|sc1|nt, 12 wg/CU", 64 * 4 * 16, 64, 12},      // 27
#endif
};
constexpr int kNumEncodeVariants = sizeof(kEncodeVariants) / sizeof(kEncodeVariants[0]);

inline unsigned grid_of(uint64_t n_tiles) { return (unsigned)(n_tiles > 0x7FFFFFFFull ? 0x7FFFFFFFull : n_tiles); }

// HIP rejects a launch whose total thread count (grid x block) exceeds 2^31-1
// ("invalid configuration argument"): 2^36 nt in 2 KiB tiles is 2^25 workgroups of
// 64 = 2^31 threads.  Large buffers are therefore cut into several launches of at
// most this many tiles (a multiple of 64, so every XCD-group permutation stays whole).
// Lab build: the tuning key "launch_tiles" lowers the limit (a multiple of 64; 0 = the hardware's) so that the tests can walk
// every launcher's several-launch loop -- and the rule that the edges ride in the LAST launch only -- at sizes of a few MiB.
#ifdef CNT_LAB_VARIANTS
inline std::atomic<int>& launch_tiles_override() {
    static std::atomic<int> v{0};
    return v;
}
#endif
inline uint64_t max_tiles_per_launch(int block) {
    const uint64_t hw = ((0x7FFFFFFFull / (uint64_t)block) / 64) * 64;
#ifdef CNT_LAB_VARIANTS
    const int o = launch_tiles_override().load(std::memory_order_relaxed);
    return o > 0 && (uint64_t)o < hw ? (uint64_t)o : hw;
#else
    return hw;
#endif
}
// how many of a launch's last workgroups share the edge items (one item per thread when there are enough tiles; the
// edge bodies are strided loops, so any count >= 1 covers all items)
constexpr unsigned kMaxEdgeGroups = 64;
inline uint32_t edge_groups(uint64_t items, unsigned block, uint64_t n_tiles) {
    uint64_t g = (items + block - 1) / block;
    if (g > kMaxEdgeGroups) g = kMaxEdgeGroups;
    if (g > n_tiles) g = n_tiles;
    return (uint32_t)g;
}

// Launches an encode: the whole tiles of [d_n, d_n + n_len) plus, in the same (last) launch, the edge words `e`
// describes (head words in front of d_n, ragged end behind the last tile).  *done_nt = nucleotides the tiles cover
// (a multiple of the variant's tile); when that is 0 NOTHING is launched and the caller runs the generic kernel.
// d_n must be 16-B aligned, d_out 4-B (16-B for the lds variant).  Returns 0 / 1 (bad variant).
// `bad` != nullptr (the *_checked entry points): variant 0's checked twin, *bad += the bytes outside the alphabet.
template <bool STRICT>
int launch_encode(int variant, const void* d_n, void* d_out, uint64_t n_len, EncodeEdges e, hipStream_t s, uint64_t* done_nt, BadCounter bad = BadCounter()) {
    if (bad) variant = 0;  // the lab's other shapes have no checked twin
    if (variant < 0 || variant >= kNumEncodeVariants) return 1;
    const uint64_t tile = kEncodeVariants[variant].tile_nt;
    const uint64_t total_tiles = n_len / tile;
    *done_nt = total_tiles * tile;
    e.tail_first = e.head_words + (*done_nt >> 5);
    const uint64_t per_launch = max_tiles_per_launch(kEncodeVariants[variant].block);
    const uint32_t xs = xcd_shift();
    const uint32_t lds = lds_for_cap(kEncodeVariants[variant].wg_cap);
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
    const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
    const bool last = first + n_tiles == total_tiles;
    const uint8_t* in = static_cast<const uint8_t*>(d_n) + first * tile;
    uint8_t* out = static_cast<uint8_t*>(d_out) + first * (tile / 4);
    e.groups = last ? edge_groups(encode_edge_items(e), kEncodeVariants[variant].block, n_tiles) : 0u;  // the edges ride in the last launch
    const dim3 g(grid_of(n_tiles));
#define CNT_ENC_STREAM(B, U, C, L, S) \
    hipLaunchKernelGGL((n_to_bits_stream<B, U, C, L, S, STRICT>), g, dim3(B), lds, s, in, out, (uint32_t)n_tiles, xs, e)
    if (bad) {
        hipLaunchKernelGGL((n_to_bits_stream_checked<64, 2, 1, kNT, kSC0 | kSC1 | kNT, STRICT>), g, dim3(64), lds, s, in, out, (uint32_t)n_tiles, xs, e, bad.p, bad.mask);
        continue;
    }
    switch (variant) {
        case 0: CNT_ENC_STREAM(64, 2, 1, kNT, kSC0 | kSC1 | kNT); break;
#ifdef CNT_LAB_VARIANTS
        case 1: CNT_ENC_STREAM(256, 1, 1, kNT, kSC1); break;
        case 2: CNT_ENC_STREAM(512, 1, 1, kSC0 | kNT, kSC1); break;
        case 3: CNT_ENC_STREAM(256, 4, 1, kNT, kNT); break;
        case 4: hipLaunchKernelGGL((n_to_bits_lds<256, 4, kNT, kSC1, STRICT>), g, dim3(256), 0, s, in, out, (uint32_t)n_tiles, e); break;
        case 5: CNT_ENC_STREAM(64, 2, 1, kNT, kSC1); break;
        case 6: CNT_ENC_STREAM(128, 2, 1, kNT, kSC1); break;
        case 7: CNT_ENC_STREAM(64, 2, 4, kNT, kSC1); break;
        case 8: CNT_ENC_STREAM(256, 4, 1, 0, 0); break;
        case 9: CNT_ENC_STREAM(64, 2, 2, kNT, kSC1); break;
        case 10: CNT_ENC_STREAM(128, 2, 1, kNT, kSC1); break;
        case 11: CNT_ENC_STREAM(64, 2, 2, kNT, kSC1); break;
        case 12: CNT_ENC_STREAM(64, 2, 2, kSC0 | kNT, kSC1); break;
        case 13: CNT_ENC_STREAM(64, 2, 2, kSC0 | kNT, kSC0 | kSC1 | kNT); break;
        case 14: CNT_ENC_STREAM(64, 2, 2, kSC1 | kNT, kSC0 | kSC1 | kNT); break;
        case 15: CNT_ENC_STREAM(64, 2, 2, kSC1, kSC0 | kSC1 | kNT); break;
        case 16: CNT_ENC_STREAM(64, 2, 2, kSC0 | kSC1 | kNT, kSC0 | kSC1 | kNT); break;
        case 17: CNT_ENC_STREAM(64, 2, 2, kNT, kSC0 | kSC1 | kNT); break;
        case 18: case 19: case 20: case 21: CNT_ENC_STREAM(64, 2, 1, kNT, kSC0 | kSC1 | kNT); break;
        case 22: CNT_ENC_STREAM(64, 2, 1, 0, kSC0 | kSC1 | kNT); break;
        case 23: CNT_ENC_STREAM(64, 2, 1, kSC1 | kNT, kSC0 | kSC1 | kNT); break;
        case 24: CNT_ENC_STREAM(64, 2, 1, kNT, kSC1 | kNT); break;
        case 25: CNT_ENC_STREAM(64, 2, 1, kNT, kSC1); break;
        case 26: CNT_ENC_STREAM(128, 2, 1, kNT, kSC0 | kSC1 | kNT); break;
        case 27: CNT_ENC_STREAM(64, 4, 1, kNT, kSC0 | kSC1 | kNT); break;
#endif
        default: return 1;
    }
    }
#undef CNT_ENC_STREAM
    return 0;
}

// The any-alignment companion of variant 0 (same shape, cache policy and residency cap):
// `base` = input pointer rounded down to 128 B, `phase` = the 1..127 bytes dropped.
constexpr int kWindowEncodeU = 4;
constexpr uint32_t kWindowEncodeTile = 64 * kWindowEncodeU * 16;
constexpr uint32_t kEncodeStreamTile = 64 * 2 * 16;  // variant 0's tile: what "a whole number of tiles" means for small inputs
constexpr uint32_t kWindowEncodeSlack = 144;  // bytes a tile may read behind its end
template <bool STRICT>
void launch_encode_window(const uint8_t* base, uint32_t phase, uint8_t* out, uint64_t total_tiles, EncodeEdges e, hipStream_t s, BadCounter bad = BadCounter()) {
    const uint64_t per_launch = max_tiles_per_launch(64);
    const uint32_t lds = std::max(lds_for_cap(12), (kWindowEncodeU + 1) * 256u);  // doubles as the kernel's exchange slab (U + 1 rows of code dwords)
    const uint32_t xs = xcd_shift();
    e.tail_first = e.head_words + total_tiles * (kWindowEncodeTile / 32);
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        e.groups = first + n_tiles == total_tiles ? edge_groups(encode_edge_items(e), 64, n_tiles) : 0u;
        if (bad)
            hipLaunchKernelGGL((n_to_bits_window_checked<kWindowEncodeU, 1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                               base + first * kWindowEncodeTile, out + first * (kWindowEncodeTile / 4), (uint32_t)n_tiles, phase, xs, e, bad.p, bad.mask);
        else
        hipLaunchKernelGGL((n_to_bits_window<kWindowEncodeU, 1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                           base + first * kWindowEncodeTile, out + first * (kWindowEncodeTile / 4), (uint32_t)n_tiles, phase, xs, e);
    }
}

// Fused round trip: whole 4-KiB tiles only; all three pointers 128-B aligned.  `cap` = resident
// one-wave workgroups per CU (0 = uncapped).  Shape from its own sweep (bench/tune_lab11.hip,
// profiles/r02_tune_lab11_*.log): one wave, FOUR 16-B loads per lane (4 KiB of ASCII in, 1 KiB of
// words + 4 KiB of ASCII out per workgroup), plain dispatch order (no XCD regrouping) and 8 resident
// workgroups per CU (72 KiB in flight per CU): 5.63-5.65 ms at 2^34 nt against 5.81-5.90 ms for
// encode's shape (two loads, XCD pairs, cap 13) that the kernel first shipped with.  Its traffic is
// 1 B read : 1.25 B written -- decode's mix, not encode's -- and like decode it wants whole 4-KiB
// pieces of the WIDE streams per workgroup.
constexpr uint32_t kRoundTripTile = 64 * 4 * 16;
constexpr uint32_t kRoundTripDefaultCap = 8;
constexpr int kRoundTripDefaultPlan = 3;  // any-alignment launch plan (device_tier.inc round_trip_plan): priced candidates
// shape 0 = the default above; shape 1 = the first shipped shape (<64, 2, 2>: two loads, XCD pairs; wants cap 13),
// two of its 2-KiB tiles per 4-KiB unit -- kept selectable (tuning key "round_trip_shape") for A/B runs
template <bool STRICT>
void launch_round_trip(const uint8_t* in, uint8_t* packed, uint8_t* back, uint64_t total_tiles, uint32_t cap, int shape, RoundTripEdges e, hipStream_t s,
                       BadCounter bad = BadCounter()) {
    if (bad) shape = 0;  // the lab's first shape has no checked twin
    // in 4-KiB units: 2^25 - 64 one-wave workgroups per launch, i.e. ONE launch up to (just under) 2^37 nt -- BASELINE.json
    // configs[3] (2^36 nt) included; the lab's shape 1 spends two workgroups per unit
    const uint64_t per_launch = max_tiles_per_launch(64) / (shape == 1 ? 2 : 1);
    const uint32_t lds = lds_for_cap(cap);
    const uint32_t xs = xcd_shift();
    e.tail_first = total_tiles * (kRoundTripTile / 32);
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        // the ragged end (< one tile: at most 128 words) rides in the last launch
        e.groups = first + n_tiles == total_tiles ? edge_groups(e.words - e.tail_first, 64, shape == 1 ? 2 * n_tiles : n_tiles) : 0u;
        const uint8_t* i0 = in + first * kRoundTripTile;
        uint8_t* p0 = packed + first * (kRoundTripTile / 4);
        uint8_t* b0 = back + first * kRoundTripTile;
#ifdef CNT_LAB_VARIANTS
        if (shape == 1) {
            hipLaunchKernelGGL((round_trip_stream<64, 2, 2, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(2 * n_tiles)), dim3(64), lds, s, i0, p0, b0, (uint32_t)(2 * n_tiles), xs, e);
            continue;
        }
#endif
        if (bad)
            hipLaunchKernelGGL((round_trip_stream_checked<64, 4, 1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s, i0, p0, b0, (uint32_t)n_tiles, xs, e, bad.p, bad.mask);
        else
        hipLaunchKernelGGL((round_trip_stream<64, 4, 1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s, i0, p0, b0, (uint32_t)n_tiles, xs, e);
    }
}

// The any-alignment fused round trip (codec2_kernels.hpp, round_trip_window): `base` = d_n + t0 rounded down to 128 B,
// the window's first byte (128-B aligned), `phase` / `phase2` = where nucleotide t0 / 16 e.p0 sit in it, `packed` = the
// (64-B aligned) address of dword e.p0, `back` = d_back + t0 (line-aligned).
// Same shape, policies and residency cap as the aligned kernel; the cap's dynamic LDS doubles as the 1280-B exchange slab.
template <bool STRICT>
void launch_round_trip_any(const uint8_t* base, uint32_t phase, uint32_t phase2, uint8_t* packed, uint8_t* back, uint64_t total_tiles, uint32_t cap,
                           RoundTripEdgesAny e, hipStream_t s, int map = 0, BadCounter bad = BadCounter()) {
    if (bad) map = 0;
    const uint64_t per_launch = max_tiles_per_launch(64);  // one workgroup per 4-KiB tile under every map: one launch up to 2^37 nt
    const uint32_t lds = std::max(lds_for_cap(cap), kRoundTripAnySlab);
    const uint32_t xs = xcd_shift();
    const uint64_t items = std::max<uint64_t>(e.p0, (e.t0 + 15) >> 4) + (e.dwords - std::min<uint64_t>(e.p1, e.t1 >> 4));
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        e.groups = first + n_tiles == total_tiles ? edge_groups(items, 64, n_tiles) : 0u;  // the edges ride in the last launch
#ifdef CNT_LAB_VARIANTS  // tuning key "round_trip_window_map": 1 = XCD pairs, 2 = XCD quads (the read-ahead of 1 / 3 of 2 / 4 tiles stays in one L2)
        if (map == 1) {
            hipLaunchKernelGGL((round_trip_window<2, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s, base + first * kRoundTripAnyTile,
                               packed + first * (kRoundTripAnyTile / 4), back + first * kRoundTripAnyTile, (uint32_t)n_tiles, phase, phase2, xs, e);
            continue;
        }
        if (map == 2) {
            hipLaunchKernelGGL((round_trip_window<4, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s, base + first * kRoundTripAnyTile,
                               packed + first * (kRoundTripAnyTile / 4), back + first * kRoundTripAnyTile, (uint32_t)n_tiles, phase, phase2, xs, e);
            continue;
        }
#else
        (void)map;
#endif
        if (bad)
            hipLaunchKernelGGL((round_trip_window_checked<1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                               base + first * kRoundTripAnyTile, packed + first * (kRoundTripAnyTile / 4), back + first * kRoundTripAnyTile,
                               (uint32_t)n_tiles, phase, phase2, xs, e, bad.p, bad.mask);
        else
        hipLaunchKernelGGL((round_trip_window<1, kNT, kSC0 | kSC1 | kNT, STRICT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                           base + first * kRoundTripAnyTile, packed + first * (kRoundTripAnyTile / 4), back + first * kRoundTripAnyTile,
                           (uint32_t)n_tiles, phase, phase2, xs, e);
    }
}

// ---- decode -------------------------------------------------------------------------
constexpr VariantDesc kDecodeVariants[] = {
    {"stream B=64 U=4 xcd-quads ld=plain st=sc0|sc1|nt, 14 wg/CU", 64 * 4 * 16, 64, 14},  // 0: default (round 3; rounds 1-2: variant 35)
#ifdef CNT_LAB_VARIANTS
    {"stream B=256 U=2 ld=plain st=sc0|sc1|nt", 256 * 2 * 16, 256, 0},        // 1
    {"stream B=64 U=2 xcd-pairs ld=plain st=sc0|sc1|nt", 64 * 2 * 16, 64, 0},  // 2
    {"stream B=256 U=2 ld=nt st=nt", 256 * 2 * 16, 256, 0},                   // 3: the first shape tried
    {"lds B=256 U=4 ld=nt st=sc0|sc1|nt", 256 * 4 * 16, 256, 0},              // 4: LDS-widened loads
    {"stream B=128 U=2 xcd-pairs ld=plain st=sc0|sc1|nt", 128 * 2 * 16, 128, 0},  // 5
    {"stream B=128 U=2 ld=nt st=sc0|sc1|nt", 128 * 2 * 16, 128, 0},           // 6
    {"stream B=128 U=2 ld=plain st=sc1|nt", 128 * 2 * 16, 128, 0},            // 7
    {"stream B=256 U=4 plain", 256 * 4 * 16, 256, 0},                         // 8: no cache-policy bits at all
    {"stream B=128 U=2 ld=plain st=sc0|sc1|nt (no residency cap)", 128 * 2 * 16, 128, 0},  // 9: as 10, uncapped
    {"stream B=128 U=2 ld=plain st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 10: as 0 without the XCD pairing
    {"stream B=128 U=2 xcd-pairs ld=nt st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 11
    {"stream B=128 U=2 xcd-pairs ld=sc0|nt st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 12
    {"stream B=128 U=2 xcd-pairs ld=plain st=sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 13
    {"stream B=128 U=2 xcd-pairs ld=sc1 st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 14
    {"stream B=128 U=2 xcd-pairs ld=plain st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 15: the default before the XCD group size was re-swept
    {"stream B=128 U=2 xcd-quads ld=plain st=sc0|sc1|nt, 14 wg/CU", 128 * 2 * 16, 128, 14},  // 16
    {"stream B=128 U=2 xcd-quads ld=plain st=sc0|sc1|nt, 15 wg/CU", 128 * 2 * 16, 128, 15},  // 17
    // round 3: one 4-KiB output page per FOUR-wave workgroup under the equivalent residency
    // (13 two-wave workgroups = 26 waves; here 6 / 7 four-wave workgroups), and the load policies not yet tried under the quad map
    {"stream B=256 U=1 xcd-quads ld=plain st=sc0|sc1|nt, 6 wg/CU", 256 * 1 * 16, 256, 6},    // 18
    {"stream B=256 U=1 xcd-quads ld=plain st=sc0|sc1|nt, 7 wg/CU", 256 * 1 * 16, 256, 7},    // 19
    {"stream B=128 U=2 xcd-quads ld=sc1 st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},    // 20
    {"stream B=128 U=2 xcd-quads ld=nt st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},     // 21
    {"stream B=128 U=2 xcd-quads ld=plain st=sc0|sc1|nt, 12 wg/CU", 128 * 2 * 16, 128, 12},  // 22
    {"stream B=128 U=2 xcd-quads ld=sc1|nt st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},   // 23
    {"stream B=128 U=2 xcd-quads ld=sc0|sc1 st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},  // 24
    {"stream B=128 U=2 xcd-quads ld=sc1 st=sc0|sc1|nt, 14 wg/CU", 128 * 2 * 16, 128, 14},      // 25
    {"stream B=128 U=2 xcd-quads ld=sc1 st=sc0|sc1|nt, 12 wg/CU", 128 * 2 * 16, 128, 12},      // 26
    // store policies under the final shape (decode is bound by the lifetime of its waves: does a store that need not
    // be acknowledged by memory retire them earlier?)
    {"stream B=128 U=2 xcd-quads ld=plain st=nt, 13 wg/CU", 128 * 2 * 16, 128, 13},            // 27
    {"stream B=128 U=2 xcd-quads ld=plain st=plain, 13 wg/CU", 128 * 2 * 16, 128, 13},         // 28
    {"stream B=128 U=2 xcd-quads ld=plain st=sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},        // 29
    {"stream B=128 U=2 xcd-quads ld=plain st=sc0|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},        // 30
    {"stream B=128 U=2 xcd-quads ld=plain st=sc1, 13 wg/CU", 128 * 2 * 16, 128, 13},           // 31
    // shapes once more, with the 4-instruction decoder: the stream is in-flight-limited (profiles/r03_bound_counters.json),
    // so: more requests per wave instead of more waves?  Yes: ONE wave per 4-KiB tile with four 4-B loads per lane, still
    // four tiles per XCD turn, 14 workgroups per CU is 0.9-1.6 % faster than two waves x two loads under 13 in encode ->
    // decode steps on six boxes out of six (profiles/r03_ab_step_decode_one_wave_tiles.log): it became variant 0, the old
    // default moved to 35 (the fused kernel's own sweep had found the same shape in round 2)
    {"stream B=128 U=4 xcd-pairs ld=plain st=sc0|sc1|nt, 7 wg/CU", 128 * 4 * 16, 128, 7},      // 32: 8-KiB tiles, 2 per XCD turn
    {"stream B=128 U=4 xcd-pairs ld=plain st=sc0|sc1|nt, 6 wg/CU", 128 * 4 * 16, 128, 6},      // 33
    {"stream B=64 U=4 xcd-quads ld=plain st=sc0|sc1|nt, 13 wg/CU", 64 * 4 * 16, 64, 13},       // 34: one wave per 4-KiB tile
    {"stream B=128 U=2 xcd-quads ld=plain st=sc0|sc1|nt, 13 wg/CU", 128 * 2 * 16, 128, 13},    // 35: the default of rounds 1-2
    {"stream B=256 U=2 xcd-pairs ld=plain st=sc0|sc1|nt, 6 wg/CU", 256 * 2 * 16, 256, 6},      // 36
    {"stream B=256 U=2 xcd-pairs ld=plain st=sc0|sc1|nt, 7 wg/CU", 256 * 2 * 16, 256, 7},      // 37
    {"stream B=64 U=4 xcd-quads ld=plain st=sc0|sc1|nt, 12 wg/CU", 64 * 4 * 16, 64, 12},       // 38
    {"stream B=64 U=4 xcd-quads ld=plain st=sc0|sc1|nt, 15 wg/CU", 64 * 4 * 16, 64, 15},       // 39
    {"stream B=64 U=4 xcd-quads ld=plain st=sc0|sc1|nt, 16 wg/CU", 64 * 4 * 16, 64, 16},       // 40
    {"stream B=64 U=4 xcd-pairs ld=plain st=sc0|sc1|nt, 14 wg/CU", 64 * 4 * 16, 64, 14},       // 41
    {"stream B=64 U=4 plain order ld=plain st=sc0|sc1|nt, 14 wg/CU", 64 * 4 * 16, 64, 14},     // 42
    {"stream B=64 U=8 xcd-pairs ld=plain st=sc0|sc1|nt, 7 wg/CU", 64 * 8 * 16, 64, 7},         // 43
    {"stream B=64 U=4 xcd-quads ld=sc1 st=sc0|sc1|nt, 14 wg/CU", 64 * 4 * 16, 64, 14},         // 44
    {"stream B=64 U=4 xcd-quads ld=plain st=sc1|nt, 14 wg/CU", 64 * 4 * 16, 64, 14},           // 45
#endif
};
constexpr int kNumDecodeVariants = sizeof(kDecodeVariants) / sizeof(kDecodeVariants[0]);

// As launch_encode: whole tiles of the output [d_out, d_out + len) plus, riding in the last launch, the edge
// nucleotides `e` describes; *done_nt == 0 means nothing was launched.
inline int launch_decode(int variant, const void* d_bits, void* d_out, uint64_t len, DecodeEdges e, hipStream_t s, uint64_t* done_nt) {
    if (variant < 0 || variant >= kNumDecodeVariants) return 1;
    const uint64_t tile = kDecodeVariants[variant].tile_nt;
    const uint64_t total_tiles = len / tile;
    *done_nt = total_tiles * tile;
    e.tail_lo = e.head + *done_nt;
    const uint64_t per_launch = max_tiles_per_launch(kDecodeVariants[variant].block);
    const uint32_t xs = xcd_shift();
    const uint32_t lds = lds_for_cap(kDecodeVariants[variant].wg_cap);
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
    const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
    const bool last = first + n_tiles == total_tiles;
    const uint8_t* in = static_cast<const uint8_t*>(d_bits) + first * (tile / 4);
    uint8_t* out = static_cast<uint8_t*>(d_out) + first * tile;
    e.groups = last ? edge_groups(decode_edge_items(e), kDecodeVariants[variant].block, n_tiles) : 0u;
    const dim3 g(grid_of(n_tiles));
    constexpr int kAll = kSC0 | kSC1 | kNT;
#define CNT_DEC_STREAM(B, U, C, L, S) \
    hipLaunchKernelGGL((bits_to_n_stream<B, U, C, L, S>), g, dim3(B), lds, s, in, out, (uint32_t)n_tiles, xs, e)
    switch (variant) {
        case 0: CNT_DEC_STREAM(64, 4, 4, 0, kAll); break;
#ifdef CNT_LAB_VARIANTS
        case 1: CNT_DEC_STREAM(256, 2, 1, 0, kAll); break;
        case 2: CNT_DEC_STREAM(64, 2, 2, 0, kAll); break;
        case 3: CNT_DEC_STREAM(256, 2, 1, kNT, kNT); break;
        case 4: hipLaunchKernelGGL((bits_to_n_lds<256, 4, kNT, kAll>), g, dim3(256), 0, s, in, out, (uint32_t)n_tiles, e); break;
        case 5: CNT_DEC_STREAM(128, 2, 2, 0, kAll); break;
        case 6: CNT_DEC_STREAM(128, 2, 1, kNT, kAll); break;
        case 7: CNT_DEC_STREAM(128, 2, 1, 0, kSC1 | kNT); break;
        case 8: CNT_DEC_STREAM(256, 4, 1, 0, 0); break;
        case 9: CNT_DEC_STREAM(128, 2, 1, 0, kAll); break;
        case 10: CNT_DEC_STREAM(128, 2, 1, 0, kAll); break;
        case 11: CNT_DEC_STREAM(128, 2, 2, kNT, kAll); break;
        case 12: CNT_DEC_STREAM(128, 2, 2, kSC0 | kNT, kAll); break;
        case 13: CNT_DEC_STREAM(128, 2, 2, 0, kSC1 | kNT); break;
        case 14: CNT_DEC_STREAM(128, 2, 2, kSC1, kAll); break;
        case 15: CNT_DEC_STREAM(128, 2, 2, 0, kAll); break;
        case 16: case 17: case 22: CNT_DEC_STREAM(128, 2, 4, 0, kAll); break;
        case 18: case 19: CNT_DEC_STREAM(256, 1, 4, 0, kAll); break;
        case 20: case 25: case 26: CNT_DEC_STREAM(128, 2, 4, kSC1, kAll); break;
        case 23: CNT_DEC_STREAM(128, 2, 4, kSC1 | kNT, kAll); break;
        case 24: CNT_DEC_STREAM(128, 2, 4, kSC0 | kSC1, kAll); break;
        case 27: CNT_DEC_STREAM(128, 2, 4, 0, kNT); break;
        case 28: CNT_DEC_STREAM(128, 2, 4, 0, 0); break;
        case 29: CNT_DEC_STREAM(128, 2, 4, 0, kSC1 | kNT); break;
        case 30: CNT_DEC_STREAM(128, 2, 4, 0, kSC0 | kNT); break;
        case 31: CNT_DEC_STREAM(128, 2, 4, 0, kSC1); break;
        case 32: case 33: CNT_DEC_STREAM(128, 4, 2, 0, kAll); break;
        case 35: CNT_DEC_STREAM(128, 2, 4, 0, kAll); break;
        case 34: case 38: case 39: case 40: CNT_DEC_STREAM(64, 4, 4, 0, kAll); break;
        case 41: CNT_DEC_STREAM(64, 4, 2, 0, kAll); break;
        case 42: CNT_DEC_STREAM(64, 4, 1, 0, kAll); break;
        case 43: CNT_DEC_STREAM(64, 8, 2, 0, kAll); break;
        case 44: CNT_DEC_STREAM(64, 4, 4, kSC1, kAll); break;
        case 45: CNT_DEC_STREAM(64, 4, 4, 0, kSC1 | kNT); break;
        case 36: case 37: CNT_DEC_STREAM(256, 2, 2, 0, kAll); break;
        case 21: CNT_DEC_STREAM(128, 2, 4, kNT, kAll); break;
#endif
        default: return 1;
    }
    }
#undef CNT_DEC_STREAM
    return 0;
}

// The any-bit-phase companion of decode variant 0 for calls INSIDE the Infinity Cache (<= 2^30 nt; larger ones take
// bits_to_n_window): `in` = the dword holding the first nucleotide, `sh` = 2 * (its index among that dword's 16).
constexpr uint32_t kShiftedDecodeTile = 64 * 4 * 16;
inline void launch_decode_shifted(const uint8_t* in, uint32_t sh, uint8_t* out, uint64_t total_tiles, DecodeEdges e, hipStream_t s) {
    const uint64_t per_launch = max_tiles_per_launch(64);
    const uint32_t lds = lds_for_cap(14);
    const uint32_t xs = xcd_shift();
    e.tail_lo = e.head + total_tiles * kShiftedDecodeTile;
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        e.groups = first + n_tiles == total_tiles ? edge_groups(decode_edge_items(e), 64, n_tiles) : 0u;
        hipLaunchKernelGGL((bits_to_n_shifted<64, 4, 4, 0, kSC0 | kSC1 | kNT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                           in + first * (kShiftedDecodeTile / 4), out + first * kShiftedDecodeTile, (uint32_t)n_tiles, sh, xs, e);
    }
}

// The window decoder (codec2_kernels.hpp, bits_to_n_window): `window` = the 128-B-aligned address at or in front of the dword
// that holds the tile sequence's first nucleotide, q = that dword's index in the window (0..31), sh = 2 * (the nucleotide's
// index among the dword's 16).  Same map, policies and residency cap as the stream kernel; the cap's dynamic LDS doubles as
// the 2-KiB slab.
inline void launch_decode_window(const uint8_t* window, uint32_t q, uint32_t sh, uint8_t* out, uint64_t total_tiles, DecodeEdges e, hipStream_t s) {
    const uint64_t per_launch = max_tiles_per_launch(64);
    const uint32_t lds = std::max(lds_for_cap(14), kWindowDecodeSlab);
    const uint32_t xs = xcd_shift();
    e.tail_lo = e.head + total_tiles * kWindowDecodeTile;
    for (uint64_t first = 0; first < total_tiles; first += per_launch) {
        const uint64_t n_tiles = total_tiles - first < per_launch ? total_tiles - first : per_launch;
        e.groups = first + n_tiles == total_tiles ? edge_groups(decode_edge_items(e), 64, n_tiles) : 0u;
        hipLaunchKernelGGL((bits_to_n_window<4, 0, kSC0 | kSC1 | kNT>), dim3(grid_of(n_tiles)), dim3(64), lds, s,
                           window + first * (kWindowDecodeTile / 4), out + first * kWindowDecodeTile, (uint32_t)n_tiles, q, sh, xs, e);
    }
}

}  // namespace cnt
