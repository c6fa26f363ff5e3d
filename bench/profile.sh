#!/bin/bash
# bench/profile.sh TAG -- rocprofv3 evidence for one round, run on the GPU box via gpurun:
#   1. --kernel-trace --stats over the headline of `python bench.py` (per-kernel durations; --no-extras keeps
#      the ceilings / configs legs out of the trace so the averages are the timed kernels' alone)
#   1b. --kernel-trace --stats over a 2^36-nt fused round trip (BASELINE.json configs[3], 144 GiB resident)
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (TCC has 4 slots: 3 + 2 do not
#      fit) over bench/pmc_workload.py (calibration probes + codec kernels)
# Raw output lands under gpurun_out/prof_TAG/ (scratch); bench/parse_profiles.py condenses it
# into profiles/ (committed).  Counters are never combined with tracing domains other than
# kernel-trace.
set -u
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
    python "$REPO/bench.py" --steps 10 --warmup 3 --cpu-seconds 0 --no-verify --no-extras > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fused36" -o fused36 -- \
    python "$REPO/bench/bench_round_trip.py" --log2-nt 36 --caps 8 --rounds 3 --iters 2 > "$OUT/fused36_under_rocprof.jsonl" 2> "$OUT/stats_fused36.err"
echo "stats fused36 rc=$?"
# 1c. the fused round trip off the 128-B grid (round_trip_window) beside the aligned kernel, 2^34 nt
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_fused_any" -o fusedany -- \
    python "$REPO/bench/align_round_trip.py" --log2-nt 34 --rounds 2 --queue 2 > "$OUT/fused_any_under_rocprof.jsonl" 2> "$OUT/stats_fused_any.err"
echo "stats fused any-alignment rc=$?"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- \
    python "$REPO/bench/pmc_workload.py" > "$OUT/pmc_fetch.log" 2>&1
echo "pmc fetch rc=$?"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- \
    python "$REPO/bench/pmc_workload.py" > "$OUT/pmc_write.log" 2>&1
echo "pmc write rc=$?"
find "$OUT" -type f | head -50
