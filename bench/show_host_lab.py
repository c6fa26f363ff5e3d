#!/usr/bin/env python3
"""one line per row of a bench/host_fresh_lab.py log: ms per 1-GiB call (encode reused, decode reused, encode fresh, decode fresh,
encode fresh freed later, decode fresh freed later), then where the caller sat"""
import json
import sys

KEYS = ("n_to_bits_hip reused", "bits_to_n_hip reused", "n_to_bits_hip fresh", "bits_to_n_hip fresh", "n_to_bits_hip fresh, freed later", "bits_to_n_hip fresh, freed later")
for l in open(sys.argv[1]):
    if not l.startswith("{"):
        continue
    j = json.loads(l)
    if "rows" not in j or not j["rows"]:
        print(j)
        continue
    r = j["rows"]
    print(j["setting"][:78].ljust(78), " ".join("%5.1f" % r[k]["ms"] for k in KEYS), r.get("caller_cpu_at_end"))
