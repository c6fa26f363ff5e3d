// probes.hip -- bandwidth-ceiling probes for DESIGN.md's roofline discussion (bench only, not
// part of the product library): same launch geometry and 16 B/lane accesses as the codec
// kernels, but no arithmetic, at the codec's read:write ratios.
//   kind 0  read-only   (xor-reduce, one conditional store per block so the loads stay live)
//   kind 1  copy 1:1
//   kind 2  read 4 : write 1   (encode's ratio, 16-B accesses both sides)
//   kind 3  read 1 : write 4   (decode's ratio)
//   kind 4  write-only
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
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
