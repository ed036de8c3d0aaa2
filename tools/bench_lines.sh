#!/bin/bash
# Bench lines of every BASELINE workload into gpurun_out/lines_<tag>/ (copy to profiles/).
TAG=$1; shift
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
for wlk in ${WORKLOADS:-c3 c2 c2_c64 c4 c5 c5_s4}; do
  extra="--no-backward"
  case $wlk in c4|c5|c5_s4) extra="$extra --cpu-images 8 --steps 30 --warmup 5 --spinup 40";; c3) extra="";; *) extra="$extra --steps 100";; esac
  timeout -k 10 400 python bench.py --workload $wlk $extra "$@" > $OUT/bench_$wlk.json 2> $OUT/bench_$wlk.err || { echo "FAILED $wlk"; tail -5 $OUT/bench_$wlk.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$wlk.json").read().strip().splitlines()[-1])
print("$wlk", round(d["value"]), "img/s", round(d["ms_per_step"],4), "ms/step lfa", round(d["lfa_ms"],4), "frac", round(d["roofline"]["frac"],4), "algo", d["roofline"]["lfa_algorithm"], "parity", d.get("parity"), "fmt", (d.get("host_format") or {}).get("end_to_end_bound"))
PY
done
