// host_tier_lab.cpp -- how should the host-pointer (drop-in) tier move data?  Times, for a
// 1 GiB ASCII buffer in ordinary pageable memory (outputs pre-touched):
//   A  the library's current host tier (cnt_n_to_bits / cnt_bits_to_n)
//   B  hipHostRegister the caller's buffers, 2-stream chunked pipeline straight from/to them
//   C  pinned staging buffers + CPU memcpy, 2-slot pipeline
//   D  one synchronous hipMemcpy each way around one kernel
// bench only.  hipcc -O2 -std=c++17 -o bench/host_tier_lab bench/host_tier_lab.cpp -Lcute_nucleotides_amd -lcute_nt_hip -Wl,-rpath,'$ORIGIN/../cute_nucleotides_amd'
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/cute_nt.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define CC(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s:%d cnt rc=%d %s\n", __FILE__, __LINE__, r_, cnt_strerror(r_)); exit(1); } } while (0)
using clk = std::chrono::steady_clock;
static double now() { return std::chrono::duration<double>(clk::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const size_t log2 = argc > 1 ? atoi(argv[1]) : 30;
    const size_t N = (size_t)1 << log2, W = N / 32;
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 64) << 20;
    std::vector<uint8_t> n(N), back(N);
    std::vector<uint64_t> bits(W), bits2(W);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < N; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; n[i] = "ACGT"[x & 3]; }
    memset(back.data(), 1, N);
    memset(bits.data(), 1, W * 8);
    memset(bits2.data(), 1, W * 8);
    void *d_in[2], *d_out[2];
    hipStream_t s[2];
    for (int i = 0; i < 2; ++i) { CK(hipMalloc(&d_in[i], N)); CK(hipMalloc(&d_out[i], N)); CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); }
    auto report = [&](const char* name, double t_enc, double t_dec) {
        printf("%-44s encode %8.2f ms = %6.2f GB/s in   decode %8.2f ms = %6.2f GB/s out\n", name, t_enc * 1e3, N / t_enc / 1e9, t_dec * 1e3, N / t_dec / 1e9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        // A: library host tier
        double t0 = now(); CC(cnt_n_to_bits(n.data(), N, bits.data(), W)); double t1 = now();
        CC(cnt_bits_to_n(bits.data(), W, N, back.data())); double t2 = now();
        if (memcmp(back.data(), n.data(), N)) { fprintf(stderr, "A mismatch\n"); return 2; }
        report("A library host tier (pageable async)", t1 - t0, t2 - t1);
        // B: register in place, 2-stream pipeline
        {
            double a0 = now();
            CK(hipHostRegister(n.data(), N, hipHostRegisterDefault)); CK(hipHostRegister(bits2.data(), W * 8, hipHostRegisterDefault));
            double a1 = now();
            int slot = 0;
            for (size_t off = 0; off < N; off += chunk, slot ^= 1) {
                const size_t m = std::min(chunk, N - off);
                CK(hipMemcpyAsync(d_in[slot], n.data() + off, m, hipMemcpyHostToDevice, s[slot]));
                CC(cnt_n_to_bits_dev(d_in[slot], m, d_out[slot], m / 32, 0, s[slot]));
                CK(hipMemcpyAsync(bits2.data() + off / 32, d_out[slot], m / 4, hipMemcpyDeviceToHost, s[slot]));
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            double a2 = now();
            CK(hipHostUnregister(n.data())); CK(hipHostUnregister(bits2.data()));
            double a3 = now();
            if (memcmp(bits2.data(), bits.data(), W * 8)) { fprintf(stderr, "B mismatch\n"); return 2; }
            printf("   B encode phases: register %.2f ms, pipeline %.2f ms (%.2f GB/s), unregister %.2f ms\n", (a1 - a0) * 1e3, (a2 - a1) * 1e3, N / (a2 - a1) / 1e9, (a3 - a2) * 1e3);
            double b0 = now();
            CK(hipHostRegister(back.data(), N, hipHostRegisterDefault)); CK(hipHostRegister(bits.data(), W * 8, hipHostRegisterDefault));
            slot = 0;
            for (size_t off = 0; off < N; off += chunk, slot ^= 1) {
                const size_t m = std::min(chunk, N - off);
                CK(hipMemcpyAsync(d_in[slot], bits.data() + off / 32, m / 4, hipMemcpyHostToDevice, s[slot]));
                CC(cnt_bits_to_n_dev(d_in[slot], m / 32, m, d_out[slot], 0, s[slot]));
                CK(hipMemcpyAsync(back.data() + off, d_out[slot], m, hipMemcpyDeviceToHost, s[slot]));
            }
            CK(hipStreamSynchronize(s[0])); CK(hipStreamSynchronize(s[1]));
            CK(hipHostUnregister(back.data())); CK(hipHostUnregister(bits.data()));
            double b1 = now();
            if (memcmp(back.data(), n.data(), N)) { fprintf(stderr, "B dec mismatch\n"); return 2; }
            report("B hipHostRegister in place + 2-stream pipeline", a3 - a0, b1 - b0);
        }
        // C: pinned staging + CPU memcpy
        {
            static uint8_t *h_in[2] = {nullptr, nullptr}, *h_out[2];
            if (!h_in[0]) for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void**)&h_in[i], chunk, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h_out[i], chunk, hipHostMallocDefault)); }
            auto run = [&](bool enc) {
                const size_t in_per_nt_num = enc ? 4 : 1, out_per_nt_num = enc ? 1 : 4;  // bytes*4 per nt
                const uint8_t* src = enc ? n.data() : (const uint8_t*)bits.data();
                uint8_t* dst = enc ? (uint8_t*)bits2.data() : back.data();
                size_t pending_off[2] = {0, 0}, pending_m[2] = {0, 0};
                int slot = 0;
                for (size_t off = 0; off < N; off += chunk, slot ^= 1) {
                    const size_t m = std::min(chunk, N - off);
                    CK(hipStreamSynchronize(s[slot]));
                    if (pending_m[slot]) memcpy(dst + pending_off[slot] * out_per_nt_num / 4, h_out[slot], pending_m[slot] * out_per_nt_num / 4);
                    memcpy(h_in[slot], src + off * in_per_nt_num / 4, m * in_per_nt_num / 4);
                    CK(hipMemcpyAsync(d_in[slot], h_in[slot], m * in_per_nt_num / 4, hipMemcpyHostToDevice, s[slot]));
                    if (enc) CC(cnt_n_to_bits_dev(d_in[slot], m, d_out[slot], m / 32, 0, s[slot]));
                    else CC(cnt_bits_to_n_dev(d_in[slot], m / 32, m, d_out[slot], 0, s[slot]));
                    CK(hipMemcpyAsync(h_out[slot], d_out[slot], m * out_per_nt_num / 4, hipMemcpyDeviceToHost, s[slot]));
                    pending_off[slot] = off; pending_m[slot] = m;
                }
                for (int k = 0; k < 2; ++k, slot ^= 1) {
                    CK(hipStreamSynchronize(s[slot]));
                    if (pending_m[slot]) memcpy(dst + pending_off[slot] * out_per_nt_num / 4, h_out[slot], pending_m[slot] * out_per_nt_num / 4);
                    pending_m[slot] = 0;
                }
            };
            double c0 = now(); run(true); double c1 = now(); run(false); double c2 = now();
            if (memcmp(bits2.data(), bits.data(), W * 8) || memcmp(back.data(), n.data(), N)) { fprintf(stderr, "C mismatch\n"); return 2; }
            report("C pinned staging + 1-thread memcpy pipeline", c1 - c0, c2 - c1);
        }
        // E: pinned staging + T-thread memcpy (fork-join per chunk)
        for (int T : {2, 4, 8}) {
            static uint8_t *h_in[2] = {nullptr, nullptr}, *h_out[2];
            if (!h_in[0]) for (int i = 0; i < 2; ++i) { CK(hipHostMalloc((void**)&h_in[i], chunk, hipHostMallocDefault)); CK(hipHostMalloc((void**)&h_out[i], chunk, hipHostMallocDefault)); }
            auto pcopy = [&](uint8_t* d, const uint8_t* sp, size_t bytes) {
                std::vector<std::thread> th;
                const size_t per = (bytes / T + 4095) / 4096 * 4096;
                for (int k = 1; k < T; ++k) { size_t lo = std::min(bytes, per * k), hi = std::min(bytes, per * (k + 1)); if (hi > lo) th.emplace_back([=] { memcpy(d + lo, sp + lo, hi - lo); }); }
                memcpy(d, sp, std::min(bytes, per));
                for (auto& t : th) t.join();
            };
            auto run = [&](bool enc) {
                const size_t in_num = enc ? 4 : 1, out_num = enc ? 1 : 4;
                const uint8_t* src = enc ? n.data() : (const uint8_t*)bits.data();
                uint8_t* dst = enc ? (uint8_t*)bits2.data() : back.data();
                size_t poff[2] = {0, 0}, pm[2] = {0, 0};
                int slot = 0;
                for (size_t off = 0; off < N; off += chunk, slot ^= 1) {
                    const size_t m = std::min(chunk, N - off);
                    CK(hipStreamSynchronize(s[slot]));
                    if (pm[slot]) pcopy(dst + poff[slot] * out_num / 4, h_out[slot], pm[slot] * out_num / 4);
                    pcopy(h_in[slot], src + off * in_num / 4, m * in_num / 4);
                    CK(hipMemcpyAsync(d_in[slot], h_in[slot], m * in_num / 4, hipMemcpyHostToDevice, s[slot]));
                    if (enc) CC(cnt_n_to_bits_dev(d_in[slot], m, d_out[slot], m / 32, 0, s[slot]));
                    else CC(cnt_bits_to_n_dev(d_in[slot], m / 32, m, d_out[slot], 0, s[slot]));
                    CK(hipMemcpyAsync(h_out[slot], d_out[slot], m * out_num / 4, hipMemcpyDeviceToHost, s[slot]));
                    poff[slot] = off; pm[slot] = m;
                }
                for (int k = 0; k < 2; ++k, slot ^= 1) { CK(hipStreamSynchronize(s[slot])); if (pm[slot]) pcopy(dst + poff[slot] * out_num / 4, h_out[slot], pm[slot] * out_num / 4); pm[slot] = 0; }
            };
            double c0 = now(); run(true); double c1 = now(); run(false); double c2 = now();
            if (memcmp(bits2.data(), bits.data(), W * 8) || memcmp(back.data(), n.data(), N)) { fprintf(stderr, "E mismatch\n"); return 2; }
            char nm[64]; snprintf(nm, sizeof nm, "E pinned staging + %d-thread memcpy pipeline", T);
            report(nm, c1 - c0, c2 - c1);
        }
        // D: synchronous whole-buffer copies
        {
            double d0 = now();
            CK(hipMemcpy(d_in[0], n.data(), N, hipMemcpyHostToDevice));
            CC(cnt_n_to_bits_dev(d_in[0], N, d_out[0], W, 0, nullptr));
            CK(hipMemcpy(bits2.data(), d_out[0], W * 8, hipMemcpyDeviceToHost));
            double d1 = now();
            CK(hipMemcpy(d_in[0], bits.data(), W * 8, hipMemcpyHostToDevice));
            CC(cnt_bits_to_n_dev(d_in[0], W, N, d_out[0], 0, nullptr));
            CK(hipMemcpy(back.data(), d_out[0], N, hipMemcpyDeviceToHost));
            double d2 = now();
            if (memcmp(bits2.data(), bits.data(), W * 8) || memcmp(back.data(), n.data(), N)) { fprintf(stderr, "D mismatch\n"); return 2; }
            report("D one sync hipMemcpy each way", d1 - d0, d2 - d1);
        }
    }
    return 0;
}
