#!/usr/bin/env python3
"""Print DESIGN.md's "round N in one table" from a bench.py JSON line, so the table is the file's numbers and
nothing else:   python bench/design_table.py profiles/r02_bench_final.json"""
import glob
import json
import os
import re
import sys

j = json.load(open(sys.argv[1]))
# The DRIVER's own measurement of the same command on its own (different) MI355X, taken at the end of a round: BENCH_rNN.json at
# the repo root, written by the driver, never by the builder.  Its `parsed` block carries the contract keys only (value,
# ms_per_step, roofline of the encode kernel), so the column has entries for those rows and a dash elsewhere.  The newest record
# is used unless one is named: `design_table.py LINE.json [BENCH_rNN.json]`.
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
drv_path = sys.argv[2] if len(sys.argv) > 2 else (sorted(glob.glob(os.path.join(ROOT, "BENCH_r[0-9][0-9].json"))) or [None])[-1]
if drv_path == "none":  # the tables of rounds 2-5 (profiles/HISTORY.md) were printed without the column
    drv_path = None
elif drv_path and not os.path.isabs(drv_path) and not os.path.exists(drv_path):
    drv_path = os.path.join(ROOT, drv_path)
drv = None
if drv_path and os.path.exists(drv_path):
    try:
        drv = json.load(open(drv_path)).get("parsed") or None
    except (OSError, ValueError):
        drv = None
drv_tag = re.sub(r"\.json$", "", os.path.basename(drv_path)) if drv else None
c = j["ceilings"]["rank0"]
tb = lambda gbs: "%.2f" % (gbs / 1000.0)
fr = lambda gbs: "%.3f" % (gbs / 8000.0)
r, d, f = j["roofline"], j["roofline_decode"], j["fused_round_trip"]
cfg = j["configs"]
k1 = cfg["configs[1] n_to_bits encode, 1 GiB (2^30 nt)"]
k2 = cfg["configs[2] bits_to_n decode, 1 GiB (2^30 nt)"]
k3 = cfg.get("configs[3] encode + decode as two passes, 64 GiB (2^36 nt)", {})
k3f = cfg.get("configs[3] fused round trip, 64 GiB (2^36 nt)", {})
s4 = j["configs4_sharded_encode"]
ev, dv = j["ceilings"]["encode_vs"], j["ceilings"]["decode_vs"]
rows = [
    ("read-only probe / write-only probe / 1 : 1 copy", "%s / %s / %s" % (tb(c["read_only"]["GBs"]), tb(c["write_only"]["GBs"]), tb(c["copy_1to1"]["GBs"])),
     "%s / %s / %s" % (fr(c["read_only"]["GBs"]), fr(c["write_only"]["GBs"]), fr(c["copy_1to1"]["GBs"])), "—"),
    ("4 : 1 probe (encode's shape) / 1 : 4 probe (decode's shape)", "%s / %s" % (tb(c["read4_write1_encode_shape"]["GBs"]), tb(c["read1_write4_decode_shape"]["GBs"])),
     "%s / %s" % (fr(c["read4_write1_encode_shape"]["GBs"]), fr(c["read1_write4_decode_shape"]["GBs"])), "—"),
    ("`n_to_bits` (mean of %d timed launches %.3f ms; median %.3f, min %.3f)" % (r["kernel_ms"]["n"], r["kernel_ms"]["mean"], r["kernel_ms"]["median"], r["kernel_ms"]["min"]),
     tb(r["achieved"]), "**%.3f**" % r["frac"], "%.3f" % ev["of_read4_write1_ceiling"]),
    ("— read-only view (the north star's \"fraction of HBM *read* bandwidth\")", tb(r["read_only_view"]["achieved"]), "**%.3f**" % r["read_only_view"]["frac"],
     "%.3f of the read-only probe" % ev["read_only_view_of_read_only_ceiling"]),
    ("`bits_to_n` (mean %.3f ms; median %.3f, min %.3f)" % (d["kernel_ms"]["mean"], d["kernel_ms"]["median"], d["kernel_ms"]["min"]), tb(d["achieved"]), "**%.3f**" % d["frac"],
     "%.3f" % dv["of_read1_write4_ceiling"]),
    ("fused round trip, 2^34 / 2^36 nt" + ("; all three pointers off the 128-B grid, 2^34" if "off_grid_frac" in f else ""),
     "%s / %s" % (tb(f["achieved"]), tb(k3f.get("achieved_GBs", 0))), "%.3f / %.3f" % (f["frac"], k3f.get("frac", 0)) + ("; %.3f" % f["off_grid_frac"] if "off_grid_frac" in f else ""),
     "%.3f of the 1 : 1 copy" % j["ceilings"]["fused_vs"]["of_copy_1to1_ceiling"] if "fused_vs" in j["ceilings"] else "—"),
    ("`configs[1]` / `[2]`: 1 GiB encode / decode", "%s / %s" % (tb(k1["achieved_GBs"]), tb(k2["achieved_GBs"])), "%.3f / %.3f" % (k1["frac"], k2["frac"]),
     "(0.2 ms launches; the 256 MiB of words sit in the Infinity Cache)"),
    ("`configs[3]`: 64 GiB, two passes", tb(k3.get("achieved_GBs", 0)), "%.3f" % k3.get("frac", 0), "—"),
    ("`configs[4]`: one GPU's 2^35-nt shard, encode", tb(1.25 * s4["per_gpu_gnts"]["min"]), "%.3f (read view %.3f)" % (s4["per_gpu_frac"]["min"], s4["per_gpu_read_view_frac"]["min"]), "—"),
]
rag = cfg.get("ragged: 2^30 - 19 nt (13 nt in the last word, zero-padded)")
if rag and "launches_per_call" in rag:  # round 3: one launch per call, 10 launches queued per event pair
    rows.append(("ragged 2^30 − 19 nt, encode / decode (ONE launch each; time relative to the aligned 2^30)", "—", "%.3f / %.3f" % (rag["encode_frac"], rag["decode_frac"]),
                 "×%.3f / ×%.3f" % (rag["encode_vs_aligned_2p30"], rag["decode_vs_aligned_2p30"])))
if "codec5" in j and "encode" in j["codec5"]:
    c5 = j["codec5"]
    rows.append(("5-letter codec (1.296 B/nt), encode / decode", "%s / %s" % (tb(c5["encode"]["achieved_GBs"]), tb(c5["decode"]["achieved_GBs"])),
                 "%.3f / %.3f" % (c5["encode"]["frac"], c5["decode"]["frac"]), "—"))
    po = j["packed_ops"]
    rows.append(("packed-domain ops: hamming / complement / reverse complement / validate", " / ".join(tb(po[k]["achieved_GBs"]) for k in ("hamming", "complement", "reverse_complement", "validate")),
                 " / ".join("%.3f" % po[k]["frac"] for k in ("hamming", "complement", "reverse_complement", "validate")), "—"))
    if "checked_encode" in po:
        ce = po["checked_encode"]
        rows.append(("validated encode `cnt_n_to_bits_checked_dev` (1.25 B/nt, one pass; validate + encode = 2.25 B/nt in two)", tb(ce["achieved_GBs"]), "**%.3f**" % ce["frac"],
                     "×%.3f of the plain encode, ×%.3f of validate + encode (same run, interleaved)" % (ce["checked_over_plain"], ce["checked_over_two_pass"])))
if drv and drv.get("roofline"):
    dr = drv["roofline"]
    extra = {2: "**%.3f** (%.3f ms)" % (dr["frac"], dr.get("avg_kernel_ms", 0)), 3: "**%.4f**" % (dr["frac"] / 1.25)}  # row index -> the driver's figure
    print("| 2^34 nt, one MI355X | TB/s | of 8 TB/s | of its own no-arithmetic ceiling | driver's box (%s), of 8 TB/s |" % drv_tag)
    print("|---|---|---|---|---|")
    for i, row in enumerate(rows):
        print("| " + " | ".join(row + (extra.get(i, "—"),)) + " |")
    print()
    print("value = %.2f Tnt/s here, %.3f Tnt/s on the driver's box (%s, %.4f ms per step); traffic %s / %s bytes per launch (%s)"
          % (j["value"] / 1000.0, drv["value"] / 1000.0, drv_tag, drv["ms_per_step"], r["traffic"], d["traffic"], (r.get("traffic_source") or "")[:40]))
else:
    print("| 2^34 nt, one MI355X | TB/s | of 8 TB/s | of its own no-arithmetic ceiling |")
    print("|---|---|---|---|")
    for row in rows:
        print("| " + " | ".join(row) + " |")
    print()
    print("value = %.2f Tnt/s; traffic %s / %s bytes per launch (%s)" % (j["value"] / 1000.0, r["traffic"], d["traffic"], (r.get("traffic_source") or "")[:40]))
cb = j.get("cpu_baseline")
if cb:
    print("cpu: all-core %.0f Gnt/s (enc %.0f, dec %.0f), one thread %.0f; cores %s" % (cb["value"], cb["encode_gnts"], cb["decode_gnts"], cb["one_thread"]["value"], cb["cores_detail"]))
    x = j["host_tier"]["crossover_vs_one_cpu_thread"]
    for k, v in x.items():
        print("crossover", k, v["host_tier_ahead_from"])
    h = j["host_tier"]
    if "pcie_ceiling" in h:
        print("pcie", h["pcie_ceiling"], {k: v for k, v in h.items() if k.startswith(("frac_of_pcie", "fresh_over"))})
