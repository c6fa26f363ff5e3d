#!/usr/bin/env python3
"""Condense gpurun_out/pmc_bound_TAG/ (bench/pmc_bound.sh) into profiles/TAG_bound_counters.json: per kernel the
mean counter values per launch, the kernel's duration under the profiler, and the derived quantities the bound
statement in profiles/TAG_encode_read_view_bound.md quotes:

  requests            TCC_EA0_RDREQ / WRREQ summed over the L2 channels (reads are 128-B requests on gfx950 for wide
                      coalesced loads: FETCH_SIZE = RDREQ x 64 B under-reports them by 2x -- MI355X_MICROARCH.md, HBM)
  avg outstanding     sum(LEVEL) / cycles-per-channel: LEVEL accumulates the number of in-flight requests every
                      cycle in every channel; cycles-per-channel = TCC_BUSY / channels (the channels are busy for the
                      whole kernel) -> chip-wide average number of requests in flight between L2 and the fabric
  avg latency         Little: outstanding / (requests / duration)
  credit-stall share  *_DRAM_CREDIT_STALL / TCC_BUSY: fraction of L2-busy cycles a channel had a request ready and
                      no credit from the memory side to send it
"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TCC_CHANNELS = 128  # 16 channels x 8 XCDs


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    log2_nt = int(sys.argv[2]) if len(sys.argv) > 2 else 34
    what = sys.argv[3] if len(sys.argv) > 3 else "codec2"
    src = os.path.join(ROOT, "gpurun_out", "pmc_bound_" + tag + ("" if what == "codec2" else "_" + what))
    n = 1 << log2_nt
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durations = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(src, "*", "pmc_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            counters[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (path, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                durations[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    want = [("read_only probe (1024 thr x 1 load, nt)", "shipped::k_read<", n, 0), ("write_only probe (256 thr x 1 store)", "shipped::k_write<", 0, n),
            ("copy 1:1 probe", "shipped::k_copy<", n, n), ("read4:write1 probe (encode's shape, no arithmetic)", "shipped::k_r4w1<64", n, n // 4),
            ("read1:write4 probe (decode's shape, no arithmetic)", "shipped::k_r1w4<", n // 4, n),
            ("read1:write4 probe + decode arithmetic x1", "shipped::k_r1w4_arith<64, 4, 4, 0, 19, 1>", n // 4, n),
            ("read1:write4 probe + decode arithmetic x2", "shipped::k_r1w4_arith<64, 4, 4, 0, 19, 2>", n // 4, n),
            ("read1:write4 probe + decode arithmetic x4", "shipped::k_r1w4_arith<64, 4, 4, 0, 19, 4>", n // 4, n),
            ("read4:write1 probe + encode arithmetic x1", "shipped::k_r4w1_arith<64, 2, 1, 2, 19, 1>", n, n // 4),
            ("read4:write1 probe + encode arithmetic x4", "shipped::k_r4w1_arith<64, 2, 1, 2, 19, 4>", n, n // 4),
            ("n_to_bits_stream (encode)", "cnt::n_to_bits_stream<", n, n // 4), ("bits_to_n_stream (decode)", "cnt::bits_to_n_stream<", n // 4, n),
            ("round_trip_stream (fused)", "cnt::round_trip_stream<", n, n + n // 4)]
    if what == "codec5":  # bench/pmc_bound_workload.py --what codec5: whole wave tiles of 128 words
        words = (n // 27) // 128 * 128
        a5, w5 = 27 * words, 8 * words
        n = a5
        want = [("5-letter encode (n_to_bits2_wave, shipped)", "cnt::n_to_bits2_wave<", a5, w5),
                ("  encode's loads + LDS staging + stores, no arithmetic", "shipped::k_enc5<1", a5, w5),
                ("  encode's loads + stores, no LDS", "shipped::k_enc5<2", a5, w5),
                ("5-letter decode (bits_to_n2_wave, shipped)", "cnt::bits_to_n2_wave<", w5, a5),
                ("  decode's loads + LDS staging + stores, no arithmetic", "shipped::k_dec5<1", w5, a5),
                ("  decode's loads + stores, no LDS", "shipped::k_dec5<2", w5, a5)]
    out = {"tag": tag, "nt": n, "tcc_channels": TCC_CHANNELS,
           "note": "bench/pmc_bound.sh over bench/pmc_bound_workload.py; counters are sums over all L2 channels, mean per launch; "
                   "durations are under the profiler (PMC passes), median over all passes", "kernels": {}}
    for label, needle, rd, wr in want:
        hits = [k for k in counters if needle in k]
        if len(hits) != 1:
            out["kernels"][label] = {"error": "expected one kernel matching %r, got %d" % (needle, len(hits))}
            continue
        k = hits[0]
        c = {name: statistics.fmean(v) for name, v in counters[k].items()}
        ms = statistics.median(durations[k])
        row = {"kernel": k[:110], "median_ms_under_pmc": round(ms, 4), "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr,
               "GBs_total": round((rd + wr) / ms / 1e6, 1), "GBs_read": round(rd / ms / 1e6, 1), "GBs_write": round(wr / ms / 1e6, 1),
               "counters_mean_per_launch": {a: round(b, 1) for a, b in sorted(c.items())}}
        d = {}
        if c.get("TCC_BUSY"):
            cyc = c["TCC_BUSY"] / TCC_CHANNELS
            d["l2_busy_cycles_per_channel"] = round(cyc)
            d["l2_clock_GHz_implied"] = round(cyc / (ms * 1e-3) / 1e9, 3)
            for side in ("RD", "WR"):
                lvl, req = c.get("TCC_EA0_%sREQ_LEVEL" % side), c.get("TCC_EA0_%sREQ" % side)
                if lvl and req:
                    outstanding = lvl / cyc
                    d["avg_outstanding_%s" % side.lower()] = round(outstanding)
                    d["avg_%s_latency_us" % side.lower()] = round(outstanding / (req / (ms * 1e-3)) * 1e6, 3)
                    d["bytes_per_%sreq" % side.lower()] = round((rd if side == "RD" else wr) / req, 1)
            for name, key in (("read_credit_stall_share_of_l2_busy", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL"),
                              ("write_credit_stall_share_of_l2_busy", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL"),
                              ("tag_stall_share_of_l2_busy", "TCC_TAG_STALL"), ("ib_stall_share_of_l2_busy", "TCC_IB_STALL")):
                if key in c:
                    d[name] = round(c[key] / c["TCC_BUSY"], 4)
        row["derived"] = d
        out["kernels"][label] = row
    dst = os.path.join(ROOT, "profiles", tag + ("_bound_counters.json" if what == "codec2" else "_bound_counters_%s.json" % what))
    json.dump(out, open(dst, "w"), indent=1)
    for label, row in out["kernels"].items():
        print(label, json.dumps({k: v for k, v in row.items() if k in ("median_ms_under_pmc", "GBs_total", "GBs_read", "derived")}))


if __name__ == "__main__":
    main()
