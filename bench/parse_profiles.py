#!/usr/bin/env python3
"""Condense gpurun_out/prof_TAG/ (raw rocprofv3 output of bench/profile.sh) into profiles/:

  profiles/TAG_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python bench.py`
  profiles/TAG_bench_under_rocprof.json   the JSON line that same run printed
  profiles/TAG_pmc_summary.json     per-kernel FETCH_SIZE / WRITE_SIZE means, calibration, HBM bytes
  profiles/hbm_traffic.json         what bench.py reports as roofline.traffic (bytes per launch)

Calibration (MI355X_MICROARCH.md section HBM): on gfx950 FETCH_SIZE under-reports wide coalesced
reads, WRITE_SIZE is uncalibrated -- so both are scaled by factors measured in the same pass on
probes with a known byte count (k_read reads exactly N bytes, k_write writes exactly N bytes).
Counters are in KiB.
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def means(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pick(d, needle):
    hits = [k for k in d if needle in k]
    if len(hits) != 1:
        raise SystemExit("expected exactly one kernel matching %r, got %r" % (needle, hits))
    return hits[0]


def traffic_summary(fetch_csv, write_csv, n, tag=""):
    """Calibrated HBM bytes per launch from one FETCH_SIZE pass and one WRITE_SIZE pass over bench/pmc_workload.py
    (n = nucleotides of the workload): the read-only / write-only probes of known size give the scale factors."""
    fetch, write = means(fetch_csv), means(write_csv)
    k_read, k_write = pick(fetch, "k_read<"), pick(write, "k_write<")
    f_cal = n / (fetch[k_read][0] * 1024.0)   # true bytes per reported FETCH_SIZE byte
    w_cal = n / (write[k_write][0] * 1024.0)
    enc, dec = pick(fetch, "n_to_bits_stream<"), pick(fetch, "bits_to_n_")  # ("n_to_bits_stream_checked<" is its own row below)

    def traffic(name, algorithmic=None):
        algorithmic = 1.25 * n if algorithmic is None else algorithmic
        rd = fetch[name][0] * 1024.0 * f_cal
        wr = write[name][0] * 1024.0 * w_cal
        return {"kernel": name, "launches": fetch[name][1], "FETCH_SIZE_KiB_mean": fetch[name][0],
                "WRITE_SIZE_KiB_mean": write[name][0], "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr),
                "hbm_bytes": int(rd + wr), "algorithmic_bytes": int(algorithmic), "ratio_to_algorithmic": round((rd + wr) / algorithmic, 4)}

    summary = {
        "tag": tag, "nt": n,
        "calibration": {"fetch_scale": round(f_cal, 4), "write_scale": round(w_cal, 4),
                        "probe_read": {"kernel": k_read, "true_bytes": n, "FETCH_SIZE_KiB_mean": fetch[k_read][0]},
                        "probe_write": {"kernel": k_write, "true_bytes": n, "WRITE_SIZE_KiB_mean": write[k_write][0]},
                        "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE); KiB units; gfx950 FETCH_SIZE reports 1/2 of wide reads"},
        "encode": traffic(enc), "decode": traffic(dec),
        "copy_probe": traffic(pick(fetch, "k_copy<"), 2.0 * n),
    }
    n5 = 27 * (1 << 28)
    for key, needle, alg in (("encode_5letter", "n_to_bits2_wave", n5 * (1 + 8 / 27)), ("decode_5letter", "bits_to_n2_wave", n5 * (1 + 8 / 27)),
                             ("hamming", "hamming_persist", 0.5 * n), ("complement", "complement_tiles", 0.5 * n),
                             ("validate", "validate_persist", 1.0 * n), ("checked_encode", "n_to_bits_stream_checked<", 1.25 * n)):
        hits = [k for k in fetch if needle in k]
        if len(hits) == 1:
            summary[key] = traffic(hits[0], alg)
    return summary


def stats_by_grid(trace_csv, out_csv):
    """rocprofv3's --stats table groups by kernel NAME only: a call that a launcher cuts into a big and a tiny launch, or a
    workload that runs the same kernel at two sizes, gets one meaningless average (round 3's fused 2^36 file: 4-us and
    22.8-ms launches of one kernel averaged to 11.4 ms).  This writes the same columns from the per-dispatch kernel trace
    grouped by (kernel, grid size), largest total first."""
    import statistics

    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace_csv)):
        groups[(r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in groups.values()) or 1
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "GridThreads", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for (name, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, grid, len(v), sum(v), round(sum(v) / len(v), 3), round(100.0 * sum(v) / total, 3), min(v), max(v),
                        round(statistics.pstdev(v), 3)])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    log2_nt = int(sys.argv[2]) if len(sys.argv) > 2 else 34
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, tag + "_bench_under_rocprof.json"))
    f36 = os.path.join(src, "stats_fused36", "fused36_kernel_stats.csv")
    if os.path.exists(f36):  # rocprofv3 --kernel-trace of a 2^36-nt fused round trip (BASELINE.json configs[3]), one row per (kernel, grid)
        stats_by_grid(os.path.join(src, "stats_fused36", "fused36_kernel_trace.csv"), os.path.join(dst, tag + "_fused_2p36_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "fused36_under_rocprof.jsonl"), os.path.join(dst, tag + "_fused_2p36_under_rocprof.jsonl"))
    fany = os.path.join(src, "stats_fused_any", "fusedany_kernel_trace.csv")
    if os.path.exists(fany):  # cnt_round_trip_dev at twelve pointer-offset triples: round_trip_window beside round_trip_stream
        stats_by_grid(fany, os.path.join(dst, tag + "_fused_any_alignment_kernel_stats.csv"))
        shutil.copy(os.path.join(src, "fused_any_under_rocprof.jsonl"), os.path.join(dst, tag + "_fused_any_alignment_under_rocprof.jsonl"))
    n = 1 << log2_nt
    summary = traffic_summary(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"),
                              os.path.join(src, "pmc_write", "pmc_counter_collection.csv"), n, tag)
    json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
    json.dump({"source": "profiles/%s_pmc_summary.json" % tag, "nt": n,
               "kernels": [summary["encode"]["kernel"].split("(")[0], summary["decode"]["kernel"].split("(")[0]],
               "encode_bytes_per_launch": summary["encode"]["hbm_bytes"],
               "decode_bytes_per_launch": summary["decode"]["hbm_bytes"]},
              open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
