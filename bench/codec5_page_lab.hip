// codec5_page_lab.hip -- would the 5-letter codec's streams run faster with PAGE tiles?  (bench only.)
// The shipped kernels tile on words: 128 words = 3456 B of ASCII = 27 lines per wave (three full 1-KiB wave accesses and
// one of 24 lanes), so every 4-KiB page of the wide stream is shared by two waves; profiles/r04_codec5_bound.md prices that
// address pattern at 6.2-6.4 TB/s against the 6.8 TB/s of the 2-bit kernels' one-page-per-wave tiles.  Here each wave owns
// ONE WHOLE 4-KiB PAGE of the wide stream (four full accesses, exactly the 2-bit shape) and the ragged piece of the narrow
// stream that belongs to it: 32768/27 = 1213.6 B of words, taken at a grain of G bytes (8 = whole words, 64 / 128 = whole
// segments / lines, which a real kernel pays for with SL extra 16-B vectors of the neighbouring page).  No arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/codec2_launch.hpp"

using namespace cnt;
typedef unsigned int vu2 __attribute__((__vector_size__(8)));

// narrow-side range of page t in units of G bytes
template <int G> __device__ __forceinline__ uint64_t unit_of(uint64_t t) { return (t * 32768ull) / (27ull * G); }

template <int C, int G, int SL, int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_enc5_page(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const uint32_t lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + t * 4096, 4096 + SL * 16);
    const uint64_t u0 = unit_of<G>(t), u1 = unit_of<G>(t + 1);
    const uint32_t nb = (uint32_t)(u1 - u0) * G;
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + u0 * G, nb);
    u32x4 v[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (i * 64 + lane) * 16, 0, LAUX));
    v[4] = u32x4{0, 0, 0, 0};
    if constexpr (SL > 0) v[4] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, lane < SL ? (256 + lane) * 16 : 0xFFFFFF00u, 0, LAUX));
    touch_residency_pad(n_tiles, v[0].x);
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { a ^= v[i].x ^ v[i].z; b ^= v[i].y ^ v[i].w; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {  // past the descriptor's range = dropped
        const vu2 w2 = {a + j, b};
        __builtin_amdgcn_raw_buffer_store_b64(w2, rout, (j * 64 + lane) * 8, 0, SAUX);
    }
}

template <int C, int G, int LAUX, int SAUX>
__global__ __launch_bounds__(64) void k_dec5_page(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint32_t n_tiles, uint32_t xs) {
    const uint64_t t = tile_of_block<C>(blockIdx.x, n_tiles, xs);
    const uint32_t lane = threadIdx.x;
    // whole units that hold any bit of this page's letters: one unit more than the page's own share at the ragged end
    const uint64_t u0 = unit_of<G>(t), u1 = unit_of<G>(t + 1) + 1;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + u0 * G, (uint32_t)(u1 - u0) * G);
    const __amdgpu_buffer_rsrc_t rout = rsrc_of(out + t * 4096, 4096);
    vu2 w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (j * 64 + lane) * 8, 0, LAUX);
    touch_residency_pad(n_tiles, w[0][0]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const u32x4 o = {w[0][0] + (uint32_t)i, w[0][1] ^ w[2][0], w[1][0] ^ w[2][1], w[1][1]};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, o), rout, (i * 64 + lane) * 16, 0, SAUX);
    }
}

// dir 0: encode's streams (a = ASCII pages, b = words) | 1: decode's (a = words, b = ASCII pages).  map = tiles per XCD turn
// (1, 4, 8).  grain = 8 / 64 / 128 (encode with 64 / 128 also reads 14 / 28 vectors of the next page; decode reads one unit
// more).  cap = workgroups per CU.  The wide buffer needs pages * 4096 + 512 bytes, the narrow one (pages + 1) * 1216 + 256.
extern "C" int page_probe(int dir, int map, int grain, int cap, const void* a, void* b, size_t pages, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint8_t* pa = static_cast<const uint8_t*>(a);
    uint8_t* pb = static_cast<uint8_t*>(b);
    if (pages == 0 || pages > 0x7FFFFFFFull / 64) return 1;
    const uint32_t t = (uint32_t)pages, xs = xcd_shift(), lds = lds_for_cap((uint32_t)cap);
    constexpr int kAll = kSC0 | kSC1 | kNT;
#define ENC(C, G, SL) hipLaunchKernelGGL((k_enc5_page<C, G, SL, kNT, kSC1>), dim3(t), dim3(64), lds, s, pa, pb, t, xs)
#define DEC(C, G) hipLaunchKernelGGL((k_dec5_page<C, G, 0, kAll>), dim3(t), dim3(64), lds, s, pa, pb, t, xs)
#define BYMAP(M, ...) do { if (map == 1) { M(1, __VA_ARGS__); } else if (map == 4) { M(4, __VA_ARGS__); } else if (map == 8) { M(8, __VA_ARGS__); } else return 1; } while (0)
    if (dir == 0) {
        if (grain == 8) BYMAP(ENC, 8, 0); else if (grain == 64) BYMAP(ENC, 64, 14); else if (grain == 128) BYMAP(ENC, 128, 28); else return 1;
    } else if (dir == 1) {
        if (grain == 8) BYMAP(DEC, 8); else if (grain == 64) BYMAP(DEC, 64); else if (grain == 128) BYMAP(DEC, 128); else return 1;
    } else return 1;
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
