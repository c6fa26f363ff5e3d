#!/bin/bash
# bench/pmc_bound.sh TAG -- counter-level evidence for where the codec kernels sit against the memory system
# (VERDICT r01 next-5c): L2 -> fabric request counts, average outstanding requests (LEVEL / busy cycles),
# credit stalls, for the shipped-shape no-arithmetic probes and the codec kernels in ONE workload.
# Counters only, each set in its own pass with --kernel-trace; bench/parse_bound.py condenses the CSVs into
# profiles/TAG_bound_counters.json.
set -u
TAG=${1:-r02}
WHAT=${2:-codec2}   # codec2 | codec5 (the second writes gpurun_out/pmc_bound_${TAG}_codec5)
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_bound_$TAG
[ "$WHAT" = codec2 ] || OUT=${OUT}_$WHAT
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in \
  "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ" \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_BUSY" \
  "TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_IB_STALL TCC_REQ" \
  "TCC_EA0_RDREQ_32B TCC_EA0_WRREQ_64B TCC_EA0_WR_UNCACHED_32B TCC_CYCLE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$i" -o pmc -- python "$REPO/bench/pmc_bound_workload.py" --what "$WHAT" > "$OUT/$i.log" 2>&1
  echo "pass $i ($set) rc=$?"
done
