#!/usr/bin/env python3
"""Condense gpurun_out/pmc_spread_TAG/ (bench/pmc_spread.sh) into profiles/TAG_decode_spread.json: per pass the 50
decode dispatches' durations with their counters, the derived clocks, and the correlation of duration with each of
them, with the output buffer (A/B), and with the launch index."""
import collections
import csv
import glob
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def corr(x, y):
    if len(x) < 3 or statistics.pstdev(x) == 0 or statistics.pstdev(y) == 0:
        return None
    mx, my = statistics.fmean(x), statistics.fmean(y)
    return round(sum((a - mx) * (b - my) for a, b in zip(x, y)) / (len(x) * statistics.pstdev(x) * statistics.pstdev(y)), 3)


def spread(v):
    s = sorted(v)
    return {"n": len(s), "min": round(s[0], 4), "p10": round(s[len(s) // 10], 4), "median": round(statistics.median(s), 4),
            "p90": round(s[(9 * len(s)) // 10], 4), "max": round(s[-1], 4), "max_over_min": round(s[-1] / s[0], 4)}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    src = os.path.join(ROOT, "gpurun_out", "pmc_spread_" + tag)
    out = {"tag": tag, "what": "bench/pmc_spread.sh: 50 back-to-back bits_to_n decodes at 2^34 nt, outputs alternate A/B", "bare_runs": [], "passes": []}
    for name in ("events_queued.jsonl", "events_gaps.jsonl"):
        path = os.path.join(src, name)
        if not os.path.exists(path):
            continue
        for line in open(path):
            line = line.strip()
            if not line.startswith("{"):
                continue
            r = json.loads(line)
            ms = r["ms"]
            out["bare_runs"].append({"file": name, "gap_us": r["gap_us"], "ms": spread(ms), "A_median": round(statistics.median(ms[0::2]), 4),
                                     "B_median": round(statistics.median(ms[1::2]), 4), "corr_with_index": corr(list(range(len(ms))), ms),
                                     "lag1_autocorr": corr(ms[:-1], ms[1:]), "series": ms})
    for path in sorted(glob.glob(os.path.join(src, "*", "pmc_counter_collection.csv"))):
        rows = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            if "bits_to_n_stream" not in r["Kernel_Name"]:
                continue
            d = rows.setdefault(int(r["Dispatch_Id"]), {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, "start": int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ds = [rows[k] for k in sorted(rows)][2:]  # the two warm-up decodes dropped
        if not ds:
            continue
        ms = [d["ms"] for d in ds]
        names = [k for k in ds[0] if k not in ("ms", "start")]
        p = {"csv": os.path.relpath(path, ROOT), "ms": spread(ms), "A_median": round(statistics.median(ms[0::2]), 4), "B_median": round(statistics.median(ms[1::2]), 4),
             "corr_ms_with_index": corr(list(range(len(ms))), ms), "counters": {}}
        for nm in names:
            v = [d.get(nm, 0.0) for d in ds]
            row = {"mean": round(statistics.fmean(v), 1), "rel_spread": round((max(v) - min(v)) / statistics.fmean(v), 5) if statistics.fmean(v) else None, "corr_with_ms": corr(v, ms)}
            if nm == "GRBM_GUI_ACTIVE":  # summed over the 8 XCDs' GRBMs? report both readings
                clk = [x / (m * 1e-3) / 1e9 for x, m in zip(v, ms)]
                row["per_ms_GHz_raw"] = spread(clk)
                row["corr_clock_with_ms"] = corr(clk, ms)
            if nm == "TCC_CYCLE":
                clk = [x / 128 / (m * 1e-3) / 1e9 for x, m in zip(v, ms)]
                row["per_channel_GHz"] = spread(clk)
                row["corr_clock_with_ms"] = corr(clk, ms)
            p["counters"][nm] = row
        # LEVEL accumulates the requests in flight every cycle, so LEVEL / REQ is the average L2->fabric latency of a request in
        # L2 cycles, per launch
        for side in ("RD", "WR"):
            lv, rq = "TCC_EA0_%sREQ_LEVEL" % side, "TCC_EA0_%sREQ" % side
            if lv in ds[0] and rq in ds[0]:
                lat = [d[lv] / d[rq] for d in ds]
                p["avg_%s_latency_cycles" % side.lower()] = dict(spread(lat), corr_with_ms=corr(lat, ms))
        p["series_ms"] = [round(x, 4) for x in ms]
        out["passes"].append(p)
    dst = os.path.join(ROOT, "profiles", tag + "_decode_spread.json")
    json.dump(out, open(dst, "w"), indent=1)
    for b in out["bare_runs"]:
        print("bare", b["file"], b["ms"], "A/B", b["A_median"], b["B_median"], "idx", b["corr_with_index"], "lag1", b["lag1_autocorr"])
    for p in out["passes"]:
        print(p["csv"], p["ms"], "A/B", p["A_median"], p["B_median"])
        for nm, row in p["counters"].items():
            print("   ", nm, row)
        for k in ("avg_rd_latency_cycles", "avg_wr_latency_cycles"):
            if k in p:
                print("   ", k, p[k])


if __name__ == "__main__":
    main()
