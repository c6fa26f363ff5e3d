#!/bin/bash
# bench/pmc_explore.sh -- exploratory PMC passes over bench/pmc_workload.py (counters only, with
# --kernel-trace; never combined with other tracing domains).  Output: gpurun_out/pmc_explore/<n>/
set -u
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out/pmc_explore
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for set in \
  "TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_BUSY" \
  "TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ_LEVEL TCC_EA0_RDREQ TCC_EA0_WRREQ" \
  "TCC_TAG_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_IB_STALL TCC_REQ" \
  "TCP_PENDING_STALL_CYCLES TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_STALL_INFLIGHT_MAX TCP_TCR_TCP_STALL_CYCLES" \
  "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/$i" -o pmc -- python "$REPO/bench/pmc_workload.py" > "$OUT/$i.log" 2>&1
  echo "pass $i ($set) rc=$?"
done
