#!/bin/bash
# bench/pmc_spread.sh TAG -- why does decode's duration vary launch to launch?
# 50 back-to-back decodes (bench/pmc_spread_workload.py), per-dispatch counters in separate --pmc passes:
#   GRBM_GUI_ACTIVE / duration = effective shader clock; TCC_CYCLE / 128 channels / duration = L2-fabric clock;
#   credit and tag stalls = the memory side pushing back.  Plus two bare runs with HIP events (queued / with gaps).
# bench/parse_spread.py condenses everything into profiles/TAG_decode_spread.json.
set -u
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_spread_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
if [ -z "${2:-}" ]; then
python "$REPO/bench/pmc_spread_workload.py" --events > "$OUT/events_queued.jsonl" 2> "$OUT/events_queued.err"
python "$REPO/bench/pmc_spread_workload.py" --events >> "$OUT/events_queued.jsonl" 2>> "$OUT/events_queued.err"
python "$REPO/bench/pmc_spread_workload.py" --events --gap-us 2000 > "$OUT/events_gaps.jsonl" 2> "$OUT/events_gaps.err"
fi
i=0
ONLY=${2:-}   # optional: run only this pass number (and skip the bare runs)
for set in \
  "GRBM_GUI_ACTIVE TCC_CYCLE TCC_BUSY TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" \
  "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY TCC_TAG_STALL TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ TCC_BUSY" \
  "GRBM_GUI_ACTIVE GRBM_COUNT TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_IB_STALL TCC_BUSY" \
  "GRBM_GUI_ACTIVE TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ"; do
  i=$((i+1))
  [ -n "$ONLY" ] && [ "$ONLY" != "$i" ] && continue
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$i" -o pmc -- python "$REPO/bench/pmc_spread_workload.py" > "$OUT/$i.log" 2>&1
  echo "pass $i ($set) rc=$?"
done
