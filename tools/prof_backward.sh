#!/bin/bash
# rocprofv3 kernel trace of tools/bench_backward.py (run through gpurun); stats to gpurun_out/prof_bwd_<tag>/
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_bwd_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$REPO
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $REPO/tools/bench_backward.py --channels-last "$@" > "$OUT/trace.log" 2>&1 || { echo FAILED; tail -5 "$OUT/trace.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print("%-70s calls %6s avg %10.1f ns  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
