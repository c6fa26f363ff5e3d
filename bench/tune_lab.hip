// tune_lab.hip -- standalone A/B lab for the 2-bit codec's memory-access shape on MI355X
// (bench only).  Same arithmetic as the product kernels (codec2_kernels.hpp), but every
// global access goes through raw buffer loads/stores so the cache-policy bits (sc0 / nt /
// sc1) of loads and stores can be chosen independently, and block size, unroll and layout
// are template knobs.  All variants run interleaved in one process; each is checked against
// the first variant's output checksum; median / min per variant are printed.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab bench/tune_lab.hip
//   bench/tune_lab [log2_nt=34] [rounds=5] [iters=2] [which=all|enc|dec]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

typedef int32_t i32x4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ u32x4 bld128(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ uint32_t bld32(__amdgpu_buffer_rsrc_t r, uint32_t off) {
    return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void bst128(__amdgpu_buffer_rsrc_t r, uint32_t off, u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), r, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void bst32(__amdgpu_buffer_rsrc_t r, uint32_t off, uint32_t v) {
    __builtin_amdgcn_raw_buffer_store_b32(v, r, off, 0, AUX);
}

// LAYOUT 0: block-interleaved (load u covers BLOCK*16 contiguous bytes), dword stores
// LAYOUT 1: wave-contiguous (a wave owns U consecutive KiB), dword stores
// LAYOUT 2: wave-contiguous + LDS transpose, 16-B stores
template <int BLOCK, int U, int LAYOUT, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void enc_lab(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_IN = BLOCK * U * 16, TILE_OUT = TILE_IN / 4;
    __shared__ __attribute__((aligned(16))) uint32_t slab[LAYOUT == 2 ? BLOCK * U : 4];
    const uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    auto rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    auto rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    u32x4 v[U];
    if constexpr (LAYOUT == 0) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = bld128<LAUX>(rin, (u * BLOCK + tid) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) bst32<SAUX>(rout, (u * BLOCK + tid) * 4, enc16<false>(v[u]));
    } else if constexpr (LAYOUT == 1) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = bld128<LAUX>(rin, ((wave * U + u) * 64 + lane) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) bst32<SAUX>(rout, ((wave * U + u) * 64 + lane) * 4, enc16<false>(v[u]));
    } else {
        static_assert(LAYOUT != 2 || U % 4 == 0, "LDS layout needs U % 4 == 0");
        uint32_t* my = slab + wave * (U * 64);
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = bld128<LAUX>(rin, ((wave * U + u) * 64 + lane) * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) my[u * 64 + lane] = enc16<false>(v[u]);
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < U / 4; ++j)
            bst128<SAUX>(rout, (wave * U * 64 + (j * 64 + lane) * 4) * 4, *reinterpret_cast<const u32x4*>(my + (j * 64 + lane) * 4));
    }
}

// decode.  LAYOUT 0: dword loads block-interleaved -> 16-B stores.  LAYOUT 1: same, wave-contiguous.
// LAYOUT 2: 16-B loads (wave-contiguous) + LDS transpose -> 16-B stores.  U counts 16-nt groups per lane.
template <int BLOCK, int U, int LAYOUT, int LAUX, int SAUX>
__global__ __launch_bounds__(BLOCK) void dec_lab(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_tiles) {
    constexpr uint32_t TILE_OUT = BLOCK * U * 16, TILE_IN = TILE_OUT / 4;
    __shared__ __attribute__((aligned(16))) uint32_t slab[LAYOUT == 2 ? BLOCK * U : 4];
    const uint64_t t = blockIdx.x;
    if (t >= n_tiles) return;
    auto rin = rsrc_of(in + t * TILE_IN, TILE_IN);
    auto rout = rsrc_of(out + t * TILE_OUT, TILE_OUT);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if constexpr (LAYOUT == 0) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = bld32<LAUX>(rin, (u * BLOCK + tid) * 4);
#pragma unroll
        for (int u = 0; u < U; ++u) bst128<SAUX>(rout, (u * BLOCK + tid) * 16, dec4(x[u]));
    } else if constexpr (LAYOUT == 1) {
        uint32_t x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = bld32<LAUX>(rin, ((wave * U + u) * 64 + lane) * 4);
#pragma unroll
        for (int u = 0; u < U; ++u) bst128<SAUX>(rout, ((wave * U + u) * 64 + lane) * 16, dec4(x[u]));
    } else {
        static_assert(LAYOUT != 2 || U % 4 == 0, "LDS layout needs U % 4 == 0");
        uint32_t* my = slab + wave * (U * 64);
        u32x4 q[U / 4 > 0 ? U / 4 : 1];
#pragma unroll
        for (int j = 0; j < U / 4; ++j) q[j] = bld128<LAUX>(rin, (wave * U * 64 + (j * 64 + lane) * 4) * 4);
#pragma unroll
        for (int j = 0; j < U / 4; ++j) *reinterpret_cast<u32x4*>(my + (j * 64 + lane) * 4) = q[j];
        wave_lds_fence();
#pragma unroll
        for (int u = 0; u < U; ++u) bst128<SAUX>(rout, ((wave * U + u) * 64 + lane) * 16, dec4(my[u * 64 + lane]));
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

template <int BLOCK, int U, int LAYOUT, int LAUX, int SAUX>
void add_enc() {
    char buf[128];
    snprintf(buf, sizeof buf, "enc B=%-4d U=%d L=%d ld=%-2d st=%-2d", BLOCK, U, LAYOUT, LAUX, SAUX);
    const uint64_t tiles = N / (uint64_t)(BLOCK * U * 16);
    variants.push_back({buf, [tiles](hipStream_t s) {
                            hipLaunchKernelGGL((enc_lab<BLOCK, U, LAYOUT, LAUX, SAUX>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, d_in, d_packed, tiles);
                        }, {}, true});
}
template <int BLOCK, int U, int LAYOUT, int LAUX, int SAUX>
void add_dec() {
    char buf[128];
    snprintf(buf, sizeof buf, "dec B=%-4d U=%d L=%d ld=%-2d st=%-2d", BLOCK, U, LAYOUT, LAUX, SAUX);
    const uint64_t tiles = N / (uint64_t)(BLOCK * U * 16);
    variants.push_back({buf, [tiles](hipStream_t s) {
                            hipLaunchKernelGGL((dec_lab<BLOCK, U, LAYOUT, LAUX, SAUX>), dim3((unsigned)tiles), dim3(BLOCK), 0, s, d_packed, d_out, tiles);
                        }, {}, false});
}

template <int LAUX>
void add_enc_staux() {
    add_enc<256, 4, 0, LAUX, 0>(); add_enc<256, 4, 0, LAUX, 1>(); add_enc<256, 4, 0, LAUX, 2>(); add_enc<256, 4, 0, LAUX, 3>();
    add_enc<256, 4, 0, LAUX, 16>(); add_enc<256, 4, 0, LAUX, 17>(); add_enc<256, 4, 0, LAUX, 18>(); add_enc<256, 4, 0, LAUX, 19>();
    add_enc<256, 4, 2, LAUX, 0>(); add_enc<256, 4, 2, LAUX, 2>(); add_enc<256, 4, 2, LAUX, 16>(); add_enc<256, 4, 2, LAUX, 18>(); add_enc<256, 4, 2, LAUX, 19>();
}
template <int LAUX>
void add_dec_staux() {
    add_dec<256, 2, 0, LAUX, 0>(); add_dec<256, 2, 0, LAUX, 1>(); add_dec<256, 2, 0, LAUX, 2>(); add_dec<256, 2, 0, LAUX, 3>();
    add_dec<256, 2, 0, LAUX, 16>(); add_dec<256, 2, 0, LAUX, 17>(); add_dec<256, 2, 0, LAUX, 18>(); add_dec<256, 2, 0, LAUX, 19>();
}
template <int BLOCK>
void add_shapes() {
    add_enc<BLOCK, 2, 0, 2, 2>(); add_enc<BLOCK, 4, 0, 2, 2>(); add_enc<BLOCK, 8, 0, 2, 2>();
    add_enc<BLOCK, 2, 1, 2, 2>(); add_enc<BLOCK, 4, 1, 2, 2>(); add_enc<BLOCK, 8, 1, 2, 2>();
    add_enc<BLOCK, 4, 2, 2, 2>(); add_enc<BLOCK, 8, 2, 2, 2>();
    add_dec<BLOCK, 1, 0, 2, 2>(); add_dec<BLOCK, 2, 0, 2, 2>(); add_dec<BLOCK, 4, 0, 2, 2>();
    add_dec<BLOCK, 1, 1, 2, 2>(); add_dec<BLOCK, 2, 1, 2, 2>(); add_dec<BLOCK, 4, 1, 2, 2>();
    add_dec<BLOCK, 4, 2, 2, 2>(); add_dec<BLOCK, 8, 2, 2, 2>();
}

template <int BLOCK, int U>
void add_policies() {
    add_enc<BLOCK, U, 0, 2, 2>(); add_enc<BLOCK, U, 0, 2, 16>(); add_enc<BLOCK, U, 0, 2, 0>(); add_enc<BLOCK, U, 0, 0, 0>();
    add_enc<BLOCK, U, 0, 3, 16>(); add_enc<BLOCK, U, 0, 2, 18>(); add_enc<BLOCK, U, 0, 2, 19>(); add_enc<BLOCK, U, 1, 2, 16>();
    add_dec<BLOCK, U, 0, 2, 2>(); add_dec<BLOCK, U, 0, 0, 19>(); add_dec<BLOCK, U, 0, 1, 19>(); add_dec<BLOCK, U, 0, 2, 19>();
    add_dec<BLOCK, U, 0, 0, 18>(); add_dec<BLOCK, U, 0, 0, 0>(); add_dec<BLOCK, U, 0, 2, 0>(); add_dec<BLOCK, U, 1, 0, 19>();
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
    const std::string which = argc > 4 ? argv[4] : "all";
    N = 1ull << log2;
    CK(hipMalloc(&d_in, N));
    CK(hipMalloc(&d_packed, N / 4));
    CK(hipMalloc(&d_out, N));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipLaunchKernelGGL(fill_random_acgt, dim3(1 << 16), dim3(kBlock), 0, s, d_in, (uint64_t)0, N, (uint64_t)0x5EED, 1);
    CK(hipStreamSynchronize(s));

    if (which == "all" || which == "enc" || which == "dec") {
        add_shapes<256>();
        add_shapes<512>();
        add_shapes<1024>();
        add_shapes<128>();
        add_enc_staux<0>(); add_enc_staux<1>(); add_enc_staux<2>(); add_enc_staux<3>();
        add_enc_staux<16>(); add_enc_staux<17>(); add_enc_staux<18>(); add_enc_staux<19>();
        add_dec_staux<0>(); add_dec_staux<1>(); add_dec_staux<2>(); add_dec_staux<3>();
        add_dec_staux<16>(); add_dec_staux<17>(); add_dec_staux<18>(); add_dec_staux<19>();
    } else {  // "small": small-workgroup shapes x the policies that mattered
        add_policies<64, 1>(); add_policies<64, 2>(); add_policies<64, 4>(); add_policies<64, 8>();
        add_policies<128, 1>(); add_policies<128, 2>(); add_policies<128, 4>();
        add_policies<256, 1>(); add_policies<256, 2>(); add_policies<256, 4>();
    }
    if (which == "enc" || which == "dec") {
        std::vector<Variant> keep;
        for (auto& v : variants)
            if ((which == "enc") == v.is_enc) keep.push_back(v);
        variants.swap(keep);
    }

    // reference results from the first variants
    uint64_t ref_enc = 0, ref_dec = 0;
    bool have_enc = false, have_dec = false;
    // make sure d_packed is valid before any decode variant runs
    hipLaunchKernelGGL((enc_lab<256, 4, 0, 0, 0>), dim3((unsigned)(N / 16384)), dim3(256), 0, s, d_in, d_packed, N / 16384);
    CK(hipStreamSynchronize(s));
    for (auto& v : variants) {
        if (v.is_enc) CK(hipMemsetAsync(d_packed, 0xFF, 4096, s)); else CK(hipMemsetAsync(d_out, 0xFF, 4096, s));
        v.launch(s);
        CK(hipGetLastError());
        uint64_t c = v.is_enc ? checksum(d_packed, N / 32, s) : checksum(d_out, N / 8, s);
        uint64_t& ref = v.is_enc ? ref_enc : ref_dec;
        bool& have = v.is_enc ? have_enc : have_dec;
        if (!have) { ref = c; have = true; }
        if (c != ref) { fprintf(stderr, "MISMATCH %s: %016llx vs %016llx\n", v.name.c_str(), (unsigned long long)c, (unsigned long long)ref); return 2; }
    }
    // decode reference must equal the input itself
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
    std::sort(variants.begin(), variants.end(), [](const Variant& a, const Variant& b) {
        auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        if (a.is_enc != b.is_enc) return a.is_enc;
        return med(a.ms) < med(b.ms);
    });
    for (auto& v : variants) {
        std::vector<float> m = v.ms;
        std::sort(m.begin(), m.end());
        const double med = m[m.size() / 2], mn = m[0];
        printf("%-36s %8.4f ms (min %8.4f)  %7.1f Gnt/s  %7.1f GB/s\n", v.name.c_str(), med, mn, N / med / 1e6, 1.25 * N / med / 1e6);
    }
    return 0;
}
