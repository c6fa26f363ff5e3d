// bench_n_to_bits.cpp -- C++ twin of the reference's criterion harness
// (benches/bench_n_to_bits.rs) for the HIP back-end: the same four groups, the same inputs
// ("ATCG" x 10000 / "ATCGN" x 8000 = 40 000 nt), the same throughput unit (input nucleotides
// = bytes per second, GiB/s) and the same rule that every timed call allocates its output
// (bench_n_to_bits.rs:6-7) -- the `*_hip` functions of cute_nucleotides.hpp return fresh
// vectors exactly like the Rust functions return fresh Vecs.  A `memcpy` row is the
// reference's comparator (:20).  A second table repeats the host-tier calls at 1 MiB ..
// 1 GiB, where the drop-in signature is PCIe-bound rather than latency-bound.
//
//   hipcc -O2 -std=c++17 -o bench/bench_n_to_bits bench/bench_n_to_bits.cpp \
//         -Lcute_nucleotides_amd -lcute_nt_hip -Wl,-rpath,'$ORIGIN/../cute_nucleotides_amd'
//
// Device-resident rows (`*_hip_dev`: the input already in HBM, enqueue + sync per call) show what is
// left when PCIe and the fresh-vector page faults are taken away -- the tier bench.py's roofline
// numbers measure.  rust/benches/bench_n_to_bits.rs is the same harness in the reference's own
// language (std-only, uncompiled here: no Rust toolchain in the image).
//
// The CPU rows of the reference's table (n_to_bits_lut, _pext, ... on the host cores) are
// produced by bench.py's cpu_baseline leg, which is the only bench allowed to call oracle/.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../cute_nucleotides_amd/cute_nucleotides.hpp"
#if __has_include(<hip/hip_runtime_api.h>)  // built by hipcc (__graft_entry__.build()): the caller's own streams and events for the ordering row
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#define CNT_TWIN_HAS_HIP 1
#endif

using namespace cute_nucleotides;
using clk = std::chrono::steady_clock;

static volatile uint64_t g_sink;

static std::vector<uint8_t> repeat(const char* unit, size_t times) {
    std::vector<uint8_t> v;
    const size_t m = strlen(unit);
    v.reserve(m * times);
    for (size_t i = 0; i < times; ++i) v.insert(v.end(), unit, unit + m);
    return v;
}

// criterion-like: warm up ~0.3 s, then time batches for ~1 s; mean per iteration
static void bench_function(const char* group, const char* name, size_t throughput_bytes, const std::function<void()>& f) {
    auto t0 = clk::now();
    size_t warm = 0;
    while (std::chrono::duration<double>(clk::now() - t0).count() < 0.3) { f(); ++warm; }
    const size_t batch = warm / 10 + 1;
    size_t iters = 0;
    t0 = clk::now();
    double el = 0;
    while ((el = std::chrono::duration<double>(clk::now() - t0).count()) < 1.0) {
        for (size_t i = 0; i < batch; ++i) f();
        iters += batch;
    }
    const double per = el / iters;
    printf("%-12s %-34s time: %10.3f us   thrpt: %9.4f GiB/s  (%.3f Gnt/s)\n", group, name, per * 1e6,
           throughput_bytes / per / (double)(1ull << 30), throughput_bytes / per / 1e9);
}

int main(int argc, char** argv) {
    const size_t max_log2 = argc > 1 ? (size_t)atoi(argv[1]) : 30;  // optional: largest host-tier size to run
    int ndev = 0;
    if (cnt_device_count(&ndev) != CNT_OK || ndev == 0) {
        fprintf(stderr, "no HIP device: this harness drives the GPU back-end only\n");
        return 1;
    }
    // ---- the reference's groups (bench_n_to_bits.rs:9-63) -----------------------------------
    {
        const auto n = repeat("ATCG", 10000);  // get_nucleotides(10000), :68-70
        bench_function("n_to_bits", "n_to_bits_hip", 40000, [&] { g_sink += n_to_bits::n_to_bits_hip(n).back(); });
        bench_function("n_to_bits", "memcpy", 40000, [&] {
            std::vector<uint8_t> dest(n.size());
            memcpy(dest.data(), n.data(), n.size());
            g_sink += dest.back();
        });
        const auto bits = n_to_bits::n_to_bits_hip(n);  // get_bits(10000), :76-78
        bench_function("bits_to_n", "bits_to_n_hip", 40000, [&] { g_sink += n_to_bits::bits_to_n_hip(bits, 40000).back(); });
        // the `_into` idiom: the caller's vector is reused, nothing is allocated or dropped inside the timed call
        Vec<uint64_t> w_into;
        Vec<uint8_t> n_into;
        bench_function("n_to_bits", "n_to_bits_hip_into", 40000, [&] { n_to_bits::n_to_bits_hip_into(n.data(), n.size(), w_into); g_sink += w_into.back(); });
        bench_function("bits_to_n", "bits_to_n_hip_into", 40000, [&] { n_to_bits::bits_to_n_hip_into(bits.data(), bits.size(), 40000, n_into); g_sink += n_into.back(); });
        if (!(w_into == bits) || !(n_into == n)) {
            fprintf(stderr, "_into mismatch at 40000 nt\n");
            return 2;
        }
        // the VALIDATED encode (round 6): the same call, and from the same pass the number of bytes BYTE_LUT would zero silently
        {
            uint64_t strays = 99;
            bench_function("n_to_bits", "n_to_bits_hip_checked", 40000, [&] { g_sink += n_to_bits::n_to_bits_hip_checked(n.data(), n.size(), strays).back(); });
            auto dirty = n;
            dirty[0] = 'N';
            dirty[20000] = '\n';
            dirty[39999] = 0;
            uint64_t strays3 = 0;
            const auto w_dirty = n_to_bits::n_to_bits_hip_checked(dirty.data(), dirty.size(), strays3);
            if (strays != 0 || strays3 != 3 || !(w_dirty == n_to_bits::n_to_bits_hip(dirty))) {
                fprintf(stderr, "checked encode mismatch at 40000 nt (%llu, %llu)\n", (unsigned long long)strays, (unsigned long long)strays3);
                return 2;
            }
        }
        // the same 40 000 nucleotides resident in HBM: one enqueue + one stream sync per call
        void *d_n = nullptr, *d_bits = nullptr, *d_back = nullptr;
        if (cnt_dev_alloc(&d_n, 40000) || cnt_dev_alloc(&d_bits, 1250 * 8) || cnt_dev_alloc(&d_back, 40000) ||
            cnt_dev_upload(d_n, n.data(), 40000)) {
            fprintf(stderr, "device allocation failed\n");
            return 1;
        }
        bench_function("n_to_bits", "n_to_bits_hip_dev (resident)", 40000, [&] {
            g_sink += (uint64_t)cnt_n_to_bits_dev(d_n, 40000, d_bits, 1250, 0, nullptr) + (uint64_t)cnt_dev_sync(nullptr);
        });
        bench_function("bits_to_n", "bits_to_n_hip_dev (resident)", 40000, [&] {
            g_sink += (uint64_t)cnt_bits_to_n_dev(d_bits, 1250, 40000, d_back, 0, nullptr) + (uint64_t)cnt_dev_sync(nullptr);
        });
        std::vector<uint64_t> chk(1250);
        std::vector<uint8_t> back(40000);
        if (cnt_dev_download(chk.data(), d_bits, 1250 * 8) || cnt_dev_download(back.data(), d_back, 40000) || chk != bits || back != n ||
            chk[0] != 0xD8D8D8D8D8D8D8D8ull) {  // the reference's unit-test vector, n_to_bits.rs:414-415
            fprintf(stderr, "device-tier mismatch at 40000 nt\n");
            return 2;
        }
        cnt_dev_free(d_n); cnt_dev_free(d_bits); cnt_dev_free(d_back);
    }
    {
        const auto n = repeat("ATCGN", 8000);  // get_nucleotides_undetermined(8000), :72-74
        bench_function("n_to_bits2", "n_to_bits2_hip", 40000, [&] { g_sink += n_to_bits2::n_to_bits2_hip(n).back(); });
        const auto bits = n_to_bits2::n_to_bits2_hip(n);
        bench_function("bits_to_n2", "bits_to_n2_hip", 40000, [&] { g_sink += n_to_bits2::bits_to_n2_hip(bits, 40000).back(); });
    }
    // ---- the same drop-in calls at sizes where PCIe, not launch latency, is the bound ---------
    for (size_t log2 : {16, 18, 20, 22, 24, 26, 30}) {
        if (log2 > max_log2) continue;
        const size_t len = (size_t)1 << log2;
        std::vector<uint8_t> n(len);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t i = 0; i < len; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;  // xorshift64
            n[i] = "ACGT"[x & 3];
        }
        char nm[64];
        snprintf(nm, sizeof nm, "n_to_bits_hip/2^%zu", log2);
        bench_function("host-tier", nm, len, [&] { g_sink += n_to_bits::n_to_bits_hip(n).back(); });
        const auto bits = n_to_bits::n_to_bits_hip(n);
        snprintf(nm, sizeof nm, "bits_to_n_hip/2^%zu", log2);
        bench_function("host-tier", nm, len, [&] { g_sink += n_to_bits::bits_to_n_hip(bits, len).back(); });
        // three timing conventions for the same call: result dropped INSIDE the timed call (the two rows above: what the
        // reference's harness does, bench_n_to_bits.rs:6-7), `_into` a vector the caller keeps (below), and the C ABI into
        // reused / fresh-malloc outputs further down
        Vec<uint64_t> w_into;
        Vec<uint8_t> n_into;
        snprintf(nm, sizeof nm, "n_to_bits_hip_into/2^%zu", log2);
        bench_function("host-tier", nm, len, [&] { n_to_bits::n_to_bits_hip_into(n.data(), len, w_into); g_sink += w_into.back(); });
        snprintf(nm, sizeof nm, "bits_to_n_hip_into/2^%zu", log2);
        bench_function("host-tier", nm, len, [&] { n_to_bits::bits_to_n_hip_into(bits.data(), bits.size(), len, n_into); g_sink += n_into.back(); });
        if (!(w_into == bits) || !(n_into == n)) {
            fprintf(stderr, "_into mismatch at 2^%zu\n", log2);
            return 2;
        }
        // the same calls through the C ABI into caller-owned, already-touched outputs: what the
        // library itself costs once the fresh-Vec page faults of the rows above are taken away
        std::vector<uint64_t> bits_out(bits.size());
        std::vector<uint8_t> n_out(len);
        snprintf(nm, sizeof nm, "cnt_n_to_bits/2^%zu (reused out)", log2);
        bench_function("host-tier", nm, len, [&] { g_sink += (uint64_t)cnt_n_to_bits(n.data(), len, bits_out.data(), bits_out.size()); });
        snprintf(nm, sizeof nm, "cnt_bits_to_n/2^%zu (reused out)", log2);
        bench_function("host-tier", nm, len, [&] { g_sink += (uint64_t)cnt_bits_to_n(bits.data(), bits.size(), len, n_out.data()); });
        if (bits_out != bits || n_out != n) {
            fprintf(stderr, "C-ABI mismatch at 2^%zu\n", log2);
            return 2;
        }
        // ... and with BOTH sides in pinned memory (PinnedBuffer = cnt_host_alloc): no staging copy either way, the copy engines
        // use the caller's buffers in place (include/cute_nt.h "pinned caller memory")
        if (log2 > 20) {
            n_to_bits::PinnedBuffer<uint8_t> pn(len), pback(len);
            n_to_bits::PinnedBuffer<uint64_t> pbits(bits.size());
            memcpy(pn.data(), n.data(), len);
            snprintf(nm, sizeof nm, "n_to_bits_hip_into/2^%zu (pinned in + out)", log2);
            bench_function("host-tier", nm, len, [&] { g_sink += n_to_bits::n_to_bits_hip_into(pn.data(), len, pbits.data(), pbits.size()); });
            snprintf(nm, sizeof nm, "bits_to_n_hip_into/2^%zu (pinned in + out)", log2);
            bench_function("host-tier", nm, len, [&] { g_sink += n_to_bits::bits_to_n_hip_into(pbits.data(), pbits.size(), len, pback.data()); });
            if (!n_to_bits::is_pinned(pn.data(), len) || memcmp(pbits.data(), bits.data(), bits.size() * 8) || memcmp(pback.data(), n.data(), len)) {
                fprintf(stderr, "pinned mismatch at 2^%zu\n", log2);
                return 2;
            }
        }
        // what the Rust wrappers do: Vec::with_capacity = malloc without zeroing, filled by the library, dropped --
        // fresh, never-touched pages on every call (the std::vector rows above also pay a single-threaded zero fill)
        snprintf(nm, sizeof nm, "cnt_n_to_bits/2^%zu (fresh malloc)", log2);
        bench_function("host-tier", nm, len, [&] {
            uint64_t* o = static_cast<uint64_t*>(malloc(bits.size() * 8));
            g_sink += (uint64_t)cnt_n_to_bits(n.data(), len, o, bits.size()) + o[bits.size() - 1];
            free(o);
        });
        snprintf(nm, sizeof nm, "cnt_bits_to_n/2^%zu (fresh malloc)", log2);
        bench_function("host-tier", nm, len, [&] {
            uint8_t* o = static_cast<uint8_t*>(malloc(len));
            g_sink += (uint64_t)cnt_bits_to_n(bits.data(), bits.size(), len, o) + o[len - 1];
            free(o);
        });
        if (n_to_bits::bits_to_n_hip(bits, len) != n) {
            fprintf(stderr, "round trip mismatch at 2^%zu\n", log2);
            return 2;
        }
        // resident rows at the same size
        void *d_n = nullptr, *d_bits = nullptr, *d_back = nullptr;
        if (cnt_dev_alloc(&d_n, len) || cnt_dev_alloc(&d_bits, bits.size() * 8) || cnt_dev_alloc(&d_back, len) || cnt_dev_upload(d_n, n.data(), len)) {
            fprintf(stderr, "device allocation failed at 2^%zu\n", log2);
            return 1;
        }
        snprintf(nm, sizeof nm, "n_to_bits_hip_dev/2^%zu (resident)", log2);
        bench_function("device-tier", nm, len, [&] { g_sink += (uint64_t)cnt_n_to_bits_dev(d_n, len, d_bits, bits.size(), 0, nullptr) + (uint64_t)cnt_dev_sync(nullptr); });
        snprintf(nm, sizeof nm, "bits_to_n_hip_dev/2^%zu (resident)", log2);
        bench_function("device-tier", nm, len, [&] { g_sink += (uint64_t)cnt_bits_to_n_dev(d_bits, bits.size(), len, d_back, 0, nullptr) + (uint64_t)cnt_dev_sync(nullptr); });
        if (cnt_dev_download(bits_out.data(), d_bits, bits.size() * 8) || cnt_dev_download(n_out.data(), d_back, len) || bits_out != bits || n_out != n) {
            fprintf(stderr, "device-tier mismatch at 2^%zu\n", log2);
            return 2;
        }
        cnt_dev_free(d_n); cnt_dev_free(d_bits); cnt_dev_free(d_back);
    }
    // ---- the enqueue-only multi-GPU device tier through the C++ mirror (one shard on this box's device 0): three
    // encode -> decode steps queued without a wait in between, one wait, timed per op ------------------------------------
    {
        const size_t len = (size_t)1 << 22;
        const auto n = repeat("ATCG", len / 4);
        device::set_device(0);
        device::DeviceBuffer d_n(n.data(), len), d_bits(len / 4), d_back(len);
        device::ShardedDevQueue q(1, /*timed=*/true);
        for (int step = 0; step < 3; ++step) {
            q.enqueue_n_to_bits({&d_n}, {len}, {&d_bits});
            q.enqueue_bits_to_n({&d_bits}, {len / 32}, {len}, {&d_back});
        }
        const auto total = q.wait();
        float sum = 0.f;
        for (size_t op = 0; op < 6; ++op) sum += q.op_ms(op)[0];
        const auto back = d_back.to_vector<uint8_t>(len);
        const auto words = d_bits.to_vector<uint64_t>(len / 32);
        bool ok = q.shards() == 1 && total[0] > 0.f && sum > 0.f && sum <= total[0] * 1.01f + 1e-3f && back == n;
        for (uint64_t w : words) ok = ok && w == 0xD8D8D8D8D8D8D8D8ull;
        if (!ok) {
            fprintf(stderr, "ShardedDevQueue mismatch\n");
            return 2;
        }
        printf("%-12s %-34s time: %10.3f us   thrpt: %9.4f GiB/s  (%.3f Gnt/s)\n", "queue", "3 x (encode + decode)/2^22, one wait", total[0] * 1e3 / 6,
               len / (total[0] * 1e-3 / 6) / (double)(1ull << 30), len / (total[0] * 1e-3 / 6) / 1e9);
    }
#ifdef CNT_TWIN_HAS_HIP
    // ---- the queue ordered against the CALLER's streams on the device (round 5), through the C++ mirror: (a) the queue adopts
    // the caller's stream -- upload, fused encode + decode and a device-side copy of the result all in order on it; (b) a
    // library-stream queue waits for the caller's event behind an asynchronous upload and records one for the caller's
    // stream to wait on.  One host synchronisation at the end of each.
    {
        const size_t len = (size_t)1 << 22;
        const auto n = repeat("ATCG", len / 4);
        device::set_device(0);
        device::DeviceBuffer d_n(len), d_bits(len / 4), d_back(len), d_copy(len);
        hipStream_t st = nullptr;
        hipEvent_t filled = nullptr, done = nullptr;
        void* pinned = nullptr;
        bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess && hipEventCreate(&filled) == hipSuccess &&
                  hipEventCreate(&done) == hipSuccess && hipHostMalloc(&pinned, len, hipHostMallocDefault) == hipSuccess;
        if (!ok) {
            fprintf(stderr, "stream / event / pinned setup failed\n");
            return 2;
        }
        memcpy(pinned, n.data(), len);
        double us = 0.0;  // the two pipelines alone (queue open .. the one synchronisation), not the checks between them
        auto t0 = std::chrono::steady_clock::now();
        {   // (a) adopted stream
            ok = ok && hipMemcpyAsync(d_n.data(), pinned, len, hipMemcpyHostToDevice, st) == hipSuccess;
            device::ShardedDevQueue q(std::vector<void*>{st});
            q.enqueue_round_trip({&d_n}, {len}, {&d_bits}, {&d_back});
            ok = ok && hipMemcpyAsync(d_copy.data(), d_back.data(), len, hipMemcpyDeviceToDevice, st) == hipSuccess;
            ok = ok && hipStreamSynchronize(st) == hipSuccess;
        }   // the queue is gone, the caller's stream is not
        us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        ok = ok && d_copy.to_vector<uint8_t>(len) == n;
        for (uint64_t w : d_bits.to_vector<uint64_t>(len / 32)) ok = ok && w == 0xD8D8D8D8D8D8D8D8ull;
        t0 = std::chrono::steady_clock::now();
        {   // (b) events
            ok = ok && hipMemsetAsync(d_n.data(), 'G', len, st) == hipSuccess && hipMemsetAsync(d_copy.data(), 0, len, st) == hipSuccess;
            ok = ok && hipEventRecord(filled, st) == hipSuccess;
            device::ShardedDevQueue q(1);
            q.wait_event(0, filled);
            q.enqueue_n_to_bits({&d_n}, {len}, {&d_bits});
            q.enqueue_bits_to_n({&d_bits}, {len / 32}, {len}, {&d_back});
            q.record_event(0, done);
            ok = ok && hipStreamWaitEvent(st, done, 0) == hipSuccess;
            ok = ok && hipMemcpyAsync(d_copy.data(), d_back.data(), len, hipMemcpyDeviceToDevice, st) == hipSuccess;
            ok = ok && hipStreamSynchronize(st) == hipSuccess;
            q.wait();
        }
        us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        ok = ok && d_copy.to_vector<uint8_t>(len) == std::vector<uint8_t>(len, (uint8_t)'G');
        for (uint64_t w : d_bits.to_vector<uint64_t>(len / 32)) ok = ok && w == ~0ull;  // G = 11
        (void)hipEventDestroy(filled); (void)hipEventDestroy(done); (void)hipHostFree(pinned); (void)hipStreamDestroy(st);
        if (!ok) {
            fprintf(stderr, "ShardedDevQueue ordering mismatch\n");
            return 2;
        }
        printf("%-12s %-34s time: %10.3f us   thrpt: %9.4f GiB/s\n", "queue", "adopted stream + events/2^22", us / 2, len / (us * 1e-6 / 2) / (double)(1ull << 30));
    }
#endif
    cnt_shutdown();
    printf("self-check ok: every C-ABI row reproduced its input (host tier, reused outputs, device tier)\n");
    return 0;
}
