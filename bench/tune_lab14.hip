// tune_lab14.hip -- can PERSISTENT waves stream at the one-shot probes' rate?  Precondition for a phase-separated
// encode (read phases of ~70 us into registers, write phases of ~18 us; profiles/r02_encode_read_view_bound.md 2a):
// a few resident waves per CU, each keeping D 1-KiB loads (or stores) in flight in a software pipeline, tiles taken
// chip-interleaved (tile = k * total_waves + wave) so that the chip-wide window stays compact.  Step 1: pure-read
// and pure-write persistent probes against the one-shot probes.  Bench only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o bench/tune_lab14 bench/tune_lab14.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/csrc/codec2_kernels.hpp"

using namespace cnt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// one wave per workgroup; wave g of G reads 1-KiB tiles g, g+G, g+2G, ...; D loads in flight
template <int D>
__global__ __launch_bounds__(64) void pread(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint64_t n_tiles) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[D];
    uint64_t t = g;
    // prologue
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const uint64_t tt = t + (uint64_t)d * G;
        const __amdgpu_buffer_rsrc_t r = rsrc_of(in + (tt < n_tiles ? tt : g) * 1024, 1024);
        buf[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, kNT));
    }
    for (; t + (uint64_t)D * G < n_tiles + (uint64_t)D * G; t += (uint64_t)D * G) {
        if (t >= n_tiles) break;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            acc ^= buf[d];  // waits for load d (in order)
            const uint64_t nx = t + (uint64_t)(D + d) * G;
            const __amdgpu_buffer_rsrc_t r = rsrc_of(in + (nx < n_tiles ? nx : g) * 1024, 1024);
            buf[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, 0, kNT));
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[lane] = acc;
}

// RUN consecutive 1-KiB tiles per step and wave (a contiguous RUN-KiB piece), D = RUN loads in flight
template <int RUN>
__global__ __launch_bounds__(64) void pread_run(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint64_t n_runs) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 buf[RUN];
    {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(in + g * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) buf[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
    }
    for (uint64_t t = g; t < n_runs; t += G) {
        const uint64_t nx = t + G < n_runs ? t + G : g;
        const __amdgpu_buffer_rsrc_t r = rsrc_of(in + nx * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) {
            acc ^= buf[d];
            buf[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[lane] = acc;
}

template <int RUN>
__global__ __launch_bounds__(64) void pwrite_run(uint8_t* __restrict__ out, uint64_t n_runs) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    u32x4 v = {(uint32_t)g, lane, 3u, 4u};
    for (uint64_t t = g; t < n_runs; t += G) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(out + t * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), r, (d * 64 + lane) * 16, 0, kSC0 | kSC1 | kNT);
        v.x += 1;
    }
}

// Step 2: the ENCODE as persistent waves with a deep software pipeline -- lab 4's persistent kernels prefetched ONE
// tile (1-2 KiB per wave) and topped out at 5.7 TB/s; the probes above say a persistent wave needs >= 8 KiB in flight.
// Wave g of G takes the contiguous RUN-KiB pieces g, g+G, ...: while it packs and stores piece t (RUN dword stores
// of 256 B each = RUN/4 KiB of consecutive output) the RUN loads of piece t+G are already in flight.
template <int RUN, int SAUX>
__global__ __launch_bounds__(64) void enc_persist_deep(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, uint64_t n_runs) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    if (g >= n_runs) return;
    u32x4 cur[RUN], nxt[RUN];
    {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(in + g * (RUN * 1024ull), RUN * 1024);
#pragma unroll
        for (int d = 0; d < RUN; ++d) cur[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
    }
    for (uint64_t t = g; t < n_runs; t += G) {
        const bool more = t + G < n_runs;
        if (more) {
            const __amdgpu_buffer_rsrc_t r = rsrc_of(in + (t + G) * (RUN * 1024ull), RUN * 1024);
#pragma unroll
            for (int d = 0; d < RUN; ++d) nxt[d] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (d * 64 + lane) * 16, 0, kNT));
        }
        const __amdgpu_buffer_rsrc_t ro = rsrc_of(out + t * (RUN * 256ull), RUN * 256);
#pragma unroll
        for (int d = 0; d < RUN; ++d) __builtin_amdgcn_raw_buffer_store_b32(enc16<false>(cur[d]), ro, (d * 64 + lane) * 4, 0, SAUX);
        if (!more) break;
#pragma unroll
        for (int d = 0; d < RUN; ++d) cur[d] = nxt[d];
    }
}

template <int D>
__global__ __launch_bounds__(64) void pwrite(uint8_t* __restrict__ out, uint64_t n_tiles) {
    const uint32_t lane = threadIdx.x;
    const uint64_t G = gridDim.x, g = blockIdx.x;
    u32x4 v = {(uint32_t)g, lane, 3u, 4u};
    for (uint64_t t = g; t < n_tiles; t += G) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(out + t * 1024, 1024);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(vu4, v), r, lane * 16, 0, kSC0 | kSC1 | kNT);
        v.x += 1;
    }
}

struct Variant { std::string name; std::function<void(hipStream_t)> launch; std::vector<float> ms; double bytes; };
static uint8_t *d_a, *d_b; static uint64_t N;
static std::vector<Variant> vs;
template <int D> void add_r(int wpc) { char n[96]; snprintf(n, 96, "persistent read  D=%-2d waves/CU=%-2d", D, wpc); const uint64_t t = N / 1024; const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((pread<D>), dim3(g), dim3(64), 0, s, d_a, d_b, t); }, {}, (double)N}); }
void add_w(int wpc) { char n[96]; snprintf(n, 96, "persistent write        waves/CU=%-2d", wpc); const uint64_t t = N / 1024; const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((pwrite<1>), dim3(g), dim3(64), 0, s, d_b, t); }, {}, (double)N}); }

template <int RUN> void add_rr(int wpc) { char n[96]; snprintf(n, 96, "persistent read  run=%-2d KiB waves/CU=%-2d", RUN, wpc); const uint64_t t = N / (RUN * 1024ull); const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((pread_run<RUN>), dim3(g), dim3(64), 0, s, d_a, d_b, t); }, {}, (double)N}); }
template <int RUN> void add_wr(int wpc) { char n[96]; snprintf(n, 96, "persistent write run=%-2d KiB waves/CU=%-2d", RUN, wpc); const uint64_t t = N / (RUN * 1024ull); const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((pwrite_run<RUN>), dim3(g), dim3(64), 0, s, d_b, t); }, {}, (double)N}); }

template <int RUN, int SAUX> void add_enc(int wpc) { char n[96]; snprintf(n, 96, "persistent ENCODE run=%-2d KiB st=%-2d waves/CU=%-2d", RUN, SAUX, wpc); const uint64_t t = N / (RUN * 1024ull); const unsigned g = 256u * wpc;
    vs.push_back({n, [t, g](hipStream_t s) { hipLaunchKernelGGL((enc_persist_deep<RUN, SAUX>), dim3(g), dim3(64), 0, s, d_a, d_b, t); }, {}, 1.25 * (double)N}); }

template <int BLOCK, int U>
__global__ __launch_bounds__(BLOCK) void oneshot_read(const uint8_t* __restrict__ in, uint8_t* __restrict__ sink, uint64_t n_tiles) {
    constexpr uint32_t TILE = BLOCK * U * 16;
    const __amdgpu_buffer_rsrc_t rin = rsrc_of(in + (uint64_t)blockIdx.x * TILE, TILE);
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rin, (u * BLOCK + threadIdx.x) * 16, 0, kNT));
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) reinterpret_cast<u32x4*>(sink)[threadIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int log2 = argc > 1 ? atoi(argv[1]) : 34, rounds = argc > 2 ? atoi(argv[2]) : 5;
    N = 1ull << log2;
    CK(hipMalloc(&d_a, N)); CK(hipMalloc(&d_b, N));
    CK(hipMemset(d_a, 0x41, N)); CK(hipMemset(d_b, 0, N));
    hipStream_t s; CK(hipStreamCreate(&s));
    { const uint64_t t = N / 16384; vs.push_back({"one-shot read 1024 thr x 1 (the probe)", [t](hipStream_t st) { hipLaunchKernelGGL((oneshot_read<1024, 1>), dim3((unsigned)t), dim3(1024), 0, st, d_a, d_b, t); }, {}, (double)N}); }
    const int set = argc > 3 ? atoi(argv[3]) : 0;
    if (set == 0) {
        for (int w : {4, 8, 12, 16, 24, 32}) { add_r<4>(w); add_r<8>(w); add_r<16>(w); }
        add_r<32>(4); add_r<32>(8);
        for (int w : {4, 8, 16, 32}) add_w(w);
    } else if (set == 2) {
        { const uint64_t t = N / 2048; vs.push_back({"one-shot ENCODE shipped (64 thr x 2, pairs, cap 23)", [t](hipStream_t st) { hipLaunchKernelGGL((n_to_bits_stream<64, 2, 2, kNT, kSC0 | kSC1 | kNT, false>), dim3((unsigned)t), dim3(64), 6912, st, d_a, d_b, t); }, {}, 1.25 * (double)N}); }
        for (int w : {2, 3, 4, 5, 6, 8, 12}) { add_enc<4, 19>(w); add_enc<8, 19>(w); add_enc<16, 19>(w); }
        for (int w : {2, 4}) { add_enc<32, 19>(w); add_enc<16, 16>(w); add_enc<8, 16>(w); }
    } else {
        for (int w : {2, 4, 6, 8, 12}) { add_rr<4>(w); add_rr<8>(w); add_rr<16>(w); add_rr<32>(w); }
        for (int w : {1, 2, 4, 6, 8, 12, 16}) { add_wr<2>(w); add_wr<4>(w); add_wr<8>(w); add_wr<16>(w); }
    }
    for (auto& v : vs) { v.launch(s); CK(hipGetLastError()); }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < rounds; ++r)
        for (auto& v : vs) {
            CK(hipEventRecord(e0, s)); v.launch(s); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); v.ms.push_back(ms);
        }
    for (auto& v : vs) { std::sort(v.ms.begin(), v.ms.end());
        printf("%-44s %8.4f ms (min %8.4f)  %7.1f GB/s\n", v.name.c_str(), (double)v.ms[v.ms.size() / 2], (double)v.ms[0], v.bytes / v.ms[v.ms.size() / 2] / 1e6); }
    return 0;
}
