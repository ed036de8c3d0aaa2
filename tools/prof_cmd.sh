#!/bin/bash
# Runs on the GPU box (through gpurun): kernel-trace stats + separate PMC passes of an
# arbitrary python command.  Usage: tools/prof_cmd.sh <tag> <script> [args...]
# (environment variables of the command are exported by the caller).
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export PYTHONPATH=$REPO
cd /tmp && export TMPDIR=/tmp
SCRIPT=$REPO/$1; shift
run() { # name, rocprof args...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 $SCRIPT $CMDARGS > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "pass $name rc=$rc"
  return $rc
}
CMDARGS="$*"
PASSES=${PASSES:-"trace fetch write sq1 sq2 l1"}
for p in $PASSES; do
  case $p in
    trace) run trace --kernel-trace --stats || exit 1;;
    fetch) run pmc_fetch --pmc FETCH_SIZE || exit 1;;
    write) run pmc_write --pmc WRITE_SIZE || exit 1;;
    l2) run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1;;
    sq1) run pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM || exit 1;;
    sq2) run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES || exit 1;;
    l1) run pmc_l1 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE || exit 1;;
  esac
done
python3 "$REPO/tools/prof_summarize.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
