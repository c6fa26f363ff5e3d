// probes.hip -- bandwidth-ceiling probes for DESIGN.md's roofline discussion (bench only, not
// part of the product library): same launch geometry and 16 B/lane accesses as the codec
// kernels, but no arithmetic, at the codec's read:write ratios.
//   kind 0  read-only   (xor-reduce, one conditional store per block so the loads stay live)
//   kind 1  copy 1:1
//   kind 2  read 4 : write 1   (encode's ratio, 16-B accesses both sides)
//   kind 3  read 1 : write 4   (decode's ratio)
//   kind 4  write-only
//
// probe_shipped(): the same five streams issued EXACTLY like the shipped codec kernels (raw buffer
// loads/stores with their cache-policy bits, the small one-shot workgroups, the XCD tile maps and the
// residency caps of codec2_launch.hpp) -- the best no-arithmetic shapes bench/tune_lab3.hip found.
// bench.py runs them in the same process as the headline so that the bench line carries same-run,
// same-box ceilings: read-only, write-only, 1:1 copy, 4:1 (encode's mix), 1:4 (decode's).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../hip/codec2_kernels.hpp"
#include "../hip/codec2_launch.hpp"
#include "../hip/codec5_kernels.hpp"

using cnt::u32x4;
constexpr int kBlock = 256;

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_read(const u32x4* __restrict__ a, u32x4* __restrict__ sink, uint64_t n_tiles) {
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + base + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x] = acc;  // practically never
}

template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_copy(const u32x4* __restrict__ a, u32x4* __restrict__ b, uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * U) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + base + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(b + base + u * kBlock, v[u]);
    }
}

// reads U*4 vectors, writes U vectors per lane per tile (tile counted in OUTPUT vectors)
template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_r4w1(const u32x4* __restrict__ a, u32x4* __restrict__ b, uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t ob = t * (uint64_t)(kBlock * U) + threadIdx.x;
        const uint64_t ib = t * (uint64_t)(kBlock * U * 4) + threadIdx.x;
        u32x4 v[U * 4];
#pragma unroll
        for (int u = 0; u < U * 4; ++u) v[u] = ld<NT>(a + ib + u * kBlock);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(b + ob + u * kBlock, v[4 * u] ^ v[4 * u + 1] ^ v[4 * u + 2] ^ v[4 * u + 3]);
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_r1w4(const u32x4* __restrict__ a, u32x4* __restrict__ b, uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t ib = t * (uint64_t)(kBlock * U) + threadIdx.x;
        const uint64_t ob = t * (uint64_t)(kBlock * U * 4) + threadIdx.x;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(a + ib + u * kBlock);
#pragma unroll
        for (int u = 0; u < U * 4; ++u) st<NT>(b + ob + u * kBlock, v[u >> 2] + (uint32_t)u);
    }
}

template <int U, bool NT>
__global__ __launch_bounds__(kBlock) void k_write(u32x4* __restrict__ b, uint64_t n_tiles) {
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const uint64_t base = t * (uint64_t)(kBlock * U) + threadIdx.x;
        u32x4 v = {(uint32_t)t, threadIdx.x, 3u, 4u};
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(b + base + u * kBlock, v);
    }
}

static unsigned grid_for(uint64_t n, int cap) {
    uint64_t g = n;
    if (cap > 0 && g > (uint64_t)cap) g = cap;
    if (g > 0x7FFFFFFFull) g = 0x7FFFFFFFull;
    return (unsigned)g;
}

// `bytes` = size of the LARGER side (the 16 B/lane side that dominates): kind 0/1/4: buffer
// size; kind 2: bytes read (a), writes bytes/4 to b; kind 3: bytes written (b), reads bytes/4.
extern "C" int probe_run(int kind, int unroll, int nt, const void* a, void* b, size_t bytes, int grid_cap, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const u32x4* pa = static_cast<const u32x4*>(a);
    u32x4* pb = static_cast<u32x4*>(b);
#define LAUNCH(K, U, NT, NTILES, ...) hipLaunchKernelGGL((K<U, NT>), dim3(grid_for(NTILES, grid_cap)), dim3(kBlock), 0, s, __VA_ARGS__, (uint64_t)(NTILES))
#define BY_NT(K, U, NTILES, ...) do { if (nt) LAUNCH(K, U, true, NTILES, __VA_ARGS__); else LAUNCH(K, U, false, NTILES, __VA_ARGS__); } while (0)
#define BY_U(K, DIV, ...) do { \
        if (unroll == 1) { uint64_t nt_ = bytes / 16 / DIV / (kBlock * 1); BY_NT(K, 1, nt_, __VA_ARGS__); } \
        else if (unroll == 2) { uint64_t nt_ = bytes / 16 / DIV / (kBlock * 2); BY_NT(K, 2, nt_, __VA_ARGS__); } \
        else if (unroll == 4) { uint64_t nt_ = bytes / 16 / DIV / (kBlock * 4); BY_NT(K, 4, nt_, __VA_ARGS__); } \
        else if (unroll == 8) { uint64_t nt_ = bytes / 16 / DIV / (kBlock * 8); BY_NT(K, 8, nt_, __VA_ARGS__); } \
        else return 1; } while (0)
    switch (kind) {
        case 0: BY_U(k_read, 1, pa, pb); break;
        case 1: BY_U(k_copy, 1, pa, pb); break;
        case 2: BY_U(k_r4w1, 4, pa, pb); break;
        case 3: BY_U(k_r1w4, 4, pa, pb); break;
        case 4: BY_U(k_write, 1, pb); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// ---- shipped-shape probes -------------------------------------------------------------------
namespace shipped {
using namespace cnt;

template <int BLOCK, int U, int LAUX>
__global__ __launch_bounds__(BLOCK) void k_read(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint32_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (uint64_t)blockIdx.x * TILE, TILE);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[threadIdx.x] = acc;  // practically never
    touch_residency_pad(n_tiles, acc.x);
}
template <int BLOCK, int U, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_write(uint8_t* __restrict__ out, uint32_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + (uint64_t)blockIdx.x * TILE, TILE);
    const u32x4 v = {(uint32_t)blockIdx.x, threadIdx.x, 3u, 4u};
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
    touch_residency_pad(n_tiles, v.x);
}
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_copy(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE, TILE), rout = rsrc_of(out + t * TILE, TILE);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v[u]), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
}
// encode's access shape without the arithmetic: 16 B in, 4 B out per lane and load
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_r4w1(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
    for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(v[u].x ^ v[u].y ^ v[u].z ^ v[u].w, rout, (u * BLOCK + threadIdx.x) * 4, 0, SAUX);
}
// decode's access shape without the arithmetic: 4 B in, 16 B out
template <int BLOCK, int U, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void k_r1w4(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    uint32_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + threadIdx.x) * 4, 0, LAUX);
    touch_residency_pad(n_tiles, x[0]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const u32x4 v = {x[u], x[u] + 1, x[u] + 2, x[u] + 3};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
    }
}
// decode's access shape WITH decode's arithmetic applied R times in a dependent chain (R = 1 is the shipped kernel's
// work; 2 and 4 ask whether the VALU time of a wave is on the critical path of this stream: the waves are resident
// under a cap, so every cycle a wave spends computing is a cycle its slot has no memory request in flight)
template <int BLOCK, int U, int C, int LAUX, int SAUX, int R>
__global__ __launch_bounds__(BLOCK) void k_r1w4_arith(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    uint32_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + threadIdx.x) * 4, 0, LAUX);
    touch_residency_pad(n_tiles, x[0]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        u32x4 v = dec4(x[u]);
#pragma unroll
        for (int r = 1; r < R; ++r) v = dec4(v.x ^ v.y ^ v.z ^ v.w);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), rout, (u * BLOCK + threadIdx.x) * 16, 0, SAUX);
    }
}
// encode's access shape with encode's arithmetic applied R times
template <int BLOCK, int U, int C, int LAUX, int SAUX, int R>
__global__ __launch_bounds__(BLOCK) void k_r4w1_arith(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_IN, TILE_IN), rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
#pragma unroll
    for (int u = 0; u < U; ++u) {
        uint32_t c = enc16<false>(v[u]);
#pragma unroll
        for (int r = 1; r < R; ++r) c = enc16<false>(u32x4{c, c ^ v[u].y, c ^ v[u].z, c ^ v[u].w});
        __builtin_amdgcn_raw_buffer_store_b32(c, rout, (u * BLOCK + threadIdx.x) * 4, 0, SAUX);
    }
}

// ---- the 5-letter codec's access patterns without its arithmetic (round 2's tune_lab16, now countable by rocprofv3) ----
// One wave per tile of 128 words: 3456 B of ASCII (27 lines: three full 1-KiB wave loads and one of 24 lanes) on the
// wide side, 1 KiB of words (two 8-B accesses per lane) on the narrow side -- exactly the shipped kernels' global
// instructions, cache policies, tile maps and residency.  MODE 1 keeps the LDS staging (16-B writes, wave fence, 8 dword
// reads per word / 6-7 dword writes per word and 16-B reads) with trivial arithmetic; MODE 2 has no LDS traffic at all.
template <int MODE, int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_enc5(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t my[kWaveDwords5 * WPL + 4];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = blockIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * TILE_BYTES, TILE_BYTES);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    constexpr int NLD = (TILE_VECS + 63) / 64;
    u32x4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        v[i] = u32x4{0, 0, 0, 0};
        if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64))
            v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    }
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    if constexpr (MODE == 1) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) *reinterpret_cast<u32x4*>(my + (i * 64 + lane) * 4) = v[i];
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const uint32_t q = (27u * lane + (uint32_t)kWaveBytes5 * j) >> 2;
            uint32_t a = 0, b = 0;
#pragma unroll
            for (int d = 0; d < 8; d += 2) { a ^= my[q + d]; b ^= my[q + d + 1]; }
            const vu2 w2 = {a, b};
            __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
        }
    } else {
        uint32_t a = 0, b = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) { a ^= v[i].x ^ v[i].z; b ^= v[i].y ^ v[i].w; }
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const vu2 w2 = {a + j, b};
            __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
        }
        if (n_tiles == ~0ull) my[lane] = a;  // keeps the static slab (and with it the residency) attached
    }
}
template <int MODE, int C, int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_dec5(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles, uint32_t xs) {
    constexpr int WPL = 2, TILE_BYTES = kWaveBytes5 * WPL, TILE_VECS = kWaveVecs5 * WPL, TILE_WORDS = kWaveWords5 * WPL;
    __shared__ __attribute__((aligned(16))) uint32_t my[kWaveDwords5 * WPL + 4];
    const uint32_t lane = threadIdx.x;
    const uint64_t t = tile_of_block<C>(blockIdx.x, (uint32_t)n_tiles, xs);
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * (TILE_WORDS * 8), TILE_WORDS * 8);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * TILE_BYTES, TILE_BYTES);
    typedef unsigned int vu2 __attribute__((__vector_size__(8)));
    vu2 w2[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) w2[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
    constexpr int NST = (TILE_VECS + 63) / 64;
    if constexpr (MODE == 1) {
        const uint32_t byte0 = 27u * lane, q0 = byte0 >> 2, cnt = ((byte0 + 27u) >> 2) - q0;
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            uint32_t* dst = my + kWaveDwords5 * j + q0;
#pragma unroll
            for (int k = 0; k < 6; ++k) dst[k] = w2[j][k & 1] + (uint32_t)k;
            if (cnt == 7) dst[6] = w2[j][0];
        }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) {
                const u32x4 o = *reinterpret_cast<const u32x4*>(my + (i * 64 + lane) * 4);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
            }
    } else {
#pragma unroll
        for (int i = 0; i < NST; ++i)
            if ((i + 1) * 64 <= TILE_VECS || lane < (uint32_t)(TILE_VECS - i * 64)) {
                const u32x4 o = {w2[0][0] + (uint32_t)i, w2[0][1], w2[1][0], w2[1][1]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
            }
        if (n_tiles == ~0ull) my[lane] = w2[0][0];
    }
}
}  // namespace shipped

// kind 0 read-only (1024 thr x 1 load, nt) | 1 copy 1:1 (256 thr x 1, ld=nt st=sc0|sc1|nt) |
// 2 read 4 : write 1 = n_to_bits_stream's shape, map, policies and residency cap | 3 read 1 : write 4 =
// bits_to_n_stream's | 4 write-only (256 thr x 1 store, sc0|sc1|nt).  `bytes` = the 16-B-per-lane side
// (kinds 0, 1, 4: the buffer; 2: bytes read from a, bytes/4 written to b; 3: bytes written to b, bytes/4
// read from a); must be a multiple of 16 KiB and at most 2^35.  Returns 0 / 1 (bad argument) / 2 (launch).
extern "C" int probe_shipped(int kind, const void* a, void* b, size_t bytes, void* stream) {
    using namespace cnt;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint8_t* pa = static_cast<const uint8_t*>(a);
    uint8_t* pb = static_cast<uint8_t*>(b);
    if (bytes == 0 || (bytes & 16383) || bytes > ((size_t)1 << 35)) return 1;
    constexpr int kAll = kSC0 | kSC1 | kNT;
    const uint32_t xs = xcd_shift();
    switch (kind) {
        case 0: { const uint32_t t = (uint32_t)(bytes / (1024 * 16)); hipLaunchKernelGGL((shipped::k_read<1024, 1, kNT>), dim3(t), dim3(1024), 0, s, pa, pb, t); break; }
        case 1: { const uint32_t t = (uint32_t)(bytes / (256 * 16)); hipLaunchKernelGGL((shipped::k_copy<256, 1, 1, kNT, kAll>), dim3(t), dim3(256), 0, s, pa, pb, t, xs); break; }
        case 2: { const uint32_t t = (uint32_t)(bytes / (64 * 2 * 16)); hipLaunchKernelGGL((shipped::k_r4w1<64, 2, 1, kNT, kAll>), dim3(t), dim3(64), lds_for_cap(23), s, pa, pb, t, xs); break; }
        case 3: { const uint32_t t = (uint32_t)(bytes / (64 * 4 * 16)); hipLaunchKernelGGL((shipped::k_r1w4<64, 4, 4, 0, kAll>), dim3(t), dim3(64), lds_for_cap(14), s, pa, pb, t, xs); break; }
        case 4: { const uint32_t t = (uint32_t)(bytes / (256 * 16)); hipLaunchKernelGGL((shipped::k_write<256, 1, kAll>), dim3(t), dim3(256), 0, s, pb, t); break; }
        // kinds 10 + R / 20 + R: the 1:4 / 4:1 shapes with the codec's own arithmetic applied R = 1, 2, 4 times (dependent chain)
#define CNT_P14(R) { const uint32_t t = (uint32_t)(bytes / (64 * 4 * 16)); hipLaunchKernelGGL((shipped::k_r1w4_arith<64, 4, 4, 0, kAll, R>), dim3(t), dim3(64), lds_for_cap(14), s, pa, pb, t, xs); break; }
#define CNT_P41(R) { const uint32_t t = (uint32_t)(bytes / (64 * 2 * 16)); hipLaunchKernelGGL((shipped::k_r4w1_arith<64, 2, 1, kNT, kAll, R>), dim3(t), dim3(64), lds_for_cap(23), s, pa, pb, t, xs); break; }
        case 11: CNT_P14(1) case 12: CNT_P14(2) case 14: CNT_P14(4) case 18: CNT_P14(8)
        case 21: CNT_P41(1) case 22: CNT_P41(2) case 24: CNT_P41(4) case 28: CNT_P41(8)
#undef CNT_P14
#undef CNT_P41
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

// The 5-letter codec's streams without its arithmetic, launched like the shipped defaults (encode2 variant 0: plain order,
// ld=nt st=sc1, 15 wg/CU; decode2 variant 0: XCD quads, ld=plain st=sc0|sc1|nt, 16 wg/CU).  kind 1 / 2: encode's pattern
// with / without the LDS staging (a = ASCII, b = words); kind 3 / 4: decode's (a = words, b = ASCII).  `words` must be a
// multiple of 128 (whole wave tiles).  Returns 0 / 1 (bad argument) / 2 (launch).
extern "C" int probe_codec5(int kind, const void* a, void* b, size_t words, void* stream) {
    using namespace cnt;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint8_t* pa = static_cast<const uint8_t*>(a);
    uint8_t* pb = static_cast<uint8_t*>(b);
    if (words == 0 || (words & 127) || (words >> 7) > 0x7FFFFFFFull / 64) return 1;
    const uint64_t tiles = words >> 7;
    constexpr int kAll = kSC0 | kSC1 | kNT;
    const uint32_t xs = xcd_shift();
    const uint32_t lds15 = lds_for_cap(15) > 3584u ? lds_for_cap(15) - 3584u : 0u, lds16 = lds_for_cap(16) > 3584u ? lds_for_cap(16) - 3584u : 0u;
    switch (kind) {
        case 1: hipLaunchKernelGGL((shipped::k_enc5<1, kNT, kSC1>), dim3((unsigned)tiles), dim3(64), lds15, s, pa, pb, tiles); break;
        case 2: hipLaunchKernelGGL((shipped::k_enc5<2, kNT, kSC1>), dim3((unsigned)tiles), dim3(64), lds15, s, pa, pb, tiles); break;
        case 3: hipLaunchKernelGGL((shipped::k_dec5<1, 4, 0, kAll>), dim3((unsigned)tiles), dim3(64), lds16, s, pa, pb, tiles, xs); break;
        case 4: hipLaunchKernelGGL((shipped::k_dec5<2, 4, 0, kAll>), dim3((unsigned)tiles), dim3(64), lds16, s, pa, pb, tiles, xs); break;
        default: return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
