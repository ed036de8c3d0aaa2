#!/bin/bash
# kernel-trace stats of one short bench run: tools/trace_one.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/trace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 170 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-backward "$@" > "$OUT/log.txt" 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/t_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "at::" in n: continue
    print("%-60s calls %5s avg_ns %10.0f" % (n[:60], r["Calls"], float(r["AverageNs"])))
PY
