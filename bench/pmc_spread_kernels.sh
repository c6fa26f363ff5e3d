#!/bin/bash
# bench/pmc_spread_kernels.sh TAG -- companion of pmc_spread.sh: the launch-to-launch spread of every stream shape
# (codec kernels and their arithmetic-free probes), 50 launches each with HIP events, bare; tells whether decode's
# spread belongs to the kernel or to the write-heavy traffic mix as such.
set -u
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_spread_$TAG
mkdir -p "$OUT"
: > "$OUT/events_kernels.jsonl"
for rep in 1 2; do
  for k in decode encode probe_r1w4 probe_write probe_read probe_r4w1 probe_copy; do
    python "$REPO/bench/pmc_spread_workload.py" --events --kernel $k >> "$OUT/events_kernels.jsonl" 2>> "$OUT/events_kernels.err"
  done
done
python - "$OUT/events_kernels.jsonl" <<'PY'
import json, statistics, sys
for line in open(sys.argv[1]):
    r = json.loads(line); ms = sorted(r["ms"])
    print("%-12s min %.4f p10 %.4f median %.4f p90 %.4f max %.4f  max/min %.4f  A %.4f B %.4f" % (r["kernel"], ms[0], ms[5], statistics.median(ms), ms[45], ms[-1], ms[-1] / ms[0],
          statistics.median(r["ms"][0::2]), statistics.median(r["ms"][1::2])))
PY
