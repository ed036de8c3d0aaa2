#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of
# one short bench.py invocation.  Usage: tools/profile.sh <tag> [bench args...]
# Raw output goes to gpurun_out/prof_<tag>/, a compact summary to
# gpurun_out/prof_<tag>/summary.txt (copy that into profiles/ to commit it).
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward $*"
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 170 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
run trace --kernel-trace --stats || exit 1
run pmc_fetch --pmc FETCH_SIZE || exit 1
run pmc_write --pmc WRITE_SIZE || exit 1
run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM || exit 1
run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES || exit 1
run pmc_l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 1
python3 "$REPO/tools/prof_summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# Back in the build container (gpurun merges gpurun_out/ only):
#   cp gpurun_out/prof_<tag>/summary.txt profiles/<tag>_rocprofv3_summary.txt
#   python tools/traffic_json.py <tag> <key, e.g. c3_bf16> 320   # LFA calls of the run: 300 spin-up + 20
