// tune_lab2.hip -- second A/B lab (bench only): fine workgroup shapes around the optimum found
// by tune_lab (few bytes per wave, many workgroups), cache-policy pairs, block->tile mappings
// and load widths, for the DIRECT layout of the 2-bit codec.  Same checking/timing protocol as
// tune_lab.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab2 bench/tune_lab2.hip
//   bench/tune_lab2 [log2_nt=34] [rounds=5] [iters=2]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"
#include "../cute_nucleotides_amd/csrc/util_kernels.hpp"

using namespace cnt;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

typedef unsigned int vu2 __attribute__((__vector_size__(8)));


// MAP 0: tile = blockIdx (dispatch order = address order; block b runs on XCD b%8)
// MAP 1: XCD-chunked: XCD k sweeps its own contiguous 1/8 of the buffer
// MAP 2: pairs: two consecutive tiles per XCD turn (tile = ((b>>4)<<4) | ((b&7)<<1) | ((b>>3)&1))
template <int MAP>
__device__ __forceinline__ uint64_t tile_of(uint64_t b, uint64_t n_tiles) {
    if constexpr (MAP == 0) return b;
    else if constexpr (MAP == 1) return (b & 7) * (n_tiles >> 3) + (b >> 3);
    else return ((b >> 4) << 4) | ((b & 7) << 1) | ((b >> 3) & 1);
}

// encode, DIRECT layout.  W = bytes loaded per lane per load (16 or 8).
template <int BLOCK, int U, int W, int MAP, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void enc2(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * W, TILE_OUT = TILE_IN / 4;
    const uint64_t t = tile_of<MAP>(blockIdx.x, n_tiles);
    auto rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    auto rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    if constexpr (W == 16) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + tid) * 16, 0, LAUX));
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(v[u]), rout, (u * BLOCK + tid) * 4, 0, SAUX);
    } else {
        vu2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b64(rin, (u * BLOCK + tid) * 8, 0, LAUX);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            uint32_t b0 = __builtin_amdgcn_ubfe(enc4<false>(v[u][0]), 19, 8), b1 = __builtin_amdgcn_ubfe(enc4<false>(v[u][1]), 19, 8);
            __builtin_amdgcn_raw_buffer_store_b16((unsigned short)(b0 | (b1 << 8)), rout, (u * BLOCK + tid) * 2, 0, SAUX);
        }
    }
}

// decode, DIRECT layout.  W = bytes loaded per lane (4 -> 16 B stored, 2 -> 8 B stored).
template <int BLOCK, int U, int W, int MAP, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void dec2(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * W, TILE_OUT = TILE_IN * 4;
    const uint64_t t = tile_of<MAP>(blockIdx.x, n_tiles);
    auto rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    auto rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x;
    if constexpr (W == 4) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rin, (u * BLOCK + tid) * 4, 0, LAUX);
#pragma unroll
        for (int u = 0; u < U; ++u) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, dec4(x[u])), rout, (u * BLOCK + tid) * 16, 0, SAUX);
    } else {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rin, (u * BLOCK + tid) * 2, 0, LAUX);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vu2 o = {dec1(x[u] & 0xFFu), dec1((x[u] >> 8) & 0xFFu)};
            __builtin_amdgcn_raw_buffer_store_b64(o, rout, (u * BLOCK + tid) * 8, 0, SAUX);
        }
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> ms;
    bool is_enc;
};
static uint8_t *d_in, *d_packed, *d_out;
static uint64_t N;
static std::vector<Variant> variants;

template <int BLOCK, int U, int W, int MAP, int LAUX, int SAUX>
void add_enc() {
    char buf[128];
    snprintf(buf, sizeof buf, "enc B=%-4d U=%d W=%-2d map=%d ld=%-2d st=%-2d", BLOCK, U, W, MAP, LAUX, SAUX);
    const uint64_t tiles = N / (uint64_t)(BLOCK * U * W);
    variants.push_back({buf, [tiles](hipStream_t s) {
                            hipLaunchKernelGGL((enc2<BLOCK, U, W, MAP, LAUX, SAUX>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, d_in, d_packed, tiles);
                        }, {}, true});
}
template <int BLOCK, int U, int W, int MAP, int LAUX, int SAUX>
void add_dec() {
    char buf[128];
    snprintf(buf, sizeof buf, "dec B=%-4d U=%d W=%-2d map=%d ld=%-2d st=%-2d", BLOCK, U, W, MAP, LAUX, SAUX);
    const uint64_t tiles = (N / 4) / (uint64_t)(BLOCK * U * W);
    variants.push_back({buf, [tiles](hipStream_t s) {
                            hipLaunchKernelGGL((dec2<BLOCK, U, W, MAP, LAUX, SAUX>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, d_packed, d_out, tiles);
                        }, {}, false});
}

template <int BLOCK, int U>
void add_enc_pol() {
    add_enc<BLOCK, U, 16, 0, 2, 16>(); add_enc<BLOCK, U, 16, 0, 2, 19>(); add_enc<BLOCK, U, 16, 0, 2, 2>(); add_enc<BLOCK, U, 16, 0, 3, 16>();
    add_enc<BLOCK, U, 16, 0, 2, 17>(); add_enc<BLOCK, U, 16, 0, 18, 16>();
}
template <int BLOCK, int U>
void add_dec_pol() {
    add_dec<BLOCK, U, 4, 0, 0, 19>(); add_dec<BLOCK, U, 4, 0, 1, 19>(); add_dec<BLOCK, U, 4, 0, 0, 18>(); add_dec<BLOCK, U, 4, 0, 2, 19>();
    add_dec<BLOCK, U, 4, 0, 0, 17>(); add_dec<BLOCK, U, 4, 0, 16, 19>();
}

static uint64_t checksum(const void* p, uint64_t words, hipStream_t s) {
    static unsigned long long* d_sum = nullptr;
    if (!d_sum) CK(hipMalloc(&d_sum, 8));
    CK(hipMemsetAsync(d_sum, 0, 8, s));
    hipLaunchKernelGGL(checksum_words, dim3(4096), dim3(kBlock), 0, s, static_cast<const uint64_t*>(p), (uint64_t)0, words, d_sum);
    unsigned long long h = 0;
    CK(hipMemcpyAsync(&h, d_sum, 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    return h;
}

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const int iters = argc > 3 ? atoi(argv[3]) : 2;
    N = 1ull << log2;
    CK(hipMalloc(&d_in, N));
    CK(hipMalloc(&d_packed, N / 4));
    CK(hipMalloc(&d_out, N));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    CK(hipStreamSynchronize(s));

    add_enc_pol<256, 1>(); add_enc_pol<512, 1>(); add_enc_pol<1024, 1>(); add_enc_pol<128, 1>(); add_enc_pol<64, 1>();
    add_enc_pol<64, 2>(); add_enc_pol<128, 2>(); add_enc_pol<256, 2>();
    add_enc<256, 1, 16, 1, 2, 16>(); add_enc<256, 1, 16, 2, 2, 16>(); add_enc<64, 2, 16, 1, 2, 16>(); add_enc<64, 2, 16, 2, 2, 16>();
    add_enc<256, 1, 8, 0, 2, 16>(); add_enc<256, 2, 8, 0, 2, 16>(); add_enc<512, 1, 8, 0, 2, 16>(); add_enc<256, 4, 8, 0, 2, 16>();
    add_dec_pol<64, 1>(); add_dec_pol<128, 1>(); add_dec_pol<256, 1>(); add_dec_pol<512, 1>();
    add_dec_pol<64, 2>(); add_dec_pol<128, 2>(); add_dec_pol<256, 2>(); add_dec_pol<128, 4>();
    add_dec<128, 2, 4, 1, 0, 19>(); add_dec<128, 2, 4, 2, 0, 19>(); add_dec<256, 1, 4, 1, 0, 19>(); add_dec<256, 1, 4, 2, 0, 19>();
    add_dec<256, 1, 2, 0, 0, 19>(); add_dec<256, 2, 2, 0, 0, 19>(); add_dec<256, 4, 2, 0, 0, 19>(); add_dec<512, 2, 2, 0, 0, 19>();

    uint64_t ref_enc = 0, ref_dec = 0;
    bool have_enc = false, have_dec = false;
    hipLaunchKernelGGL((enc2<256, 4, 16, 0, 0, 0>), dim3((unsigned)(N / 16384)), dim3(256), 0, s, d_in, d_packed, N / 16384);
    CK(hipStreamSynchronize(s));
    for (auto& v : variants) {
        if (v.is_enc) CK(hipMemsetAsync(d_packed, 0xFF, 1 << 20, s)); else CK(hipMemsetAsync(d_out, 0xFF, 1 << 20, s));
        v.launch(s);
        CK(hipGetLastError());
        uint64_t c = v.is_enc ? checksum(d_packed, N / 32, s) : checksum(d_out, N / 8, s);
        uint64_t& ref = v.is_enc ? ref_enc : ref_dec;
        bool& have = v.is_enc ? have_enc : have_dec;
        if (!have) { ref = c; have = true; }
        if (c != ref) { fprintf(stderr, "MISMATCH %s\n", v.name.c_str()); return 2; }
    }
    if (have_dec && checksum(d_in, N / 8, s) != ref_dec) { fprintf(stderr, "decode != input\n"); return 3; }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : variants) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; ++i) v.launch(s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            v.ms.push_back(ms / iters);
        }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::sort(variants.begin(), variants.end(), [&](const Variant& a, const Variant& b) {
        if (a.is_enc != b.is_enc) return a.is_enc;
        return med(a.ms) < med(b.ms);
    });
    for (auto& v : variants) {
        std::vector<float> m = v.ms;
        std::sort(m.begin(), m.end());
        printf("%-44s %8.4f ms (min %8.4f)  %7.1f Gnt/s  %7.1f GB/s\n", v.name.c_str(), (double)m[m.size() / 2], (double)m[0], N / m[m.size() / 2] / 1e6,
               1.25 * N / m[m.size() / 2] / 1e6);
    }
    return 0;
}
