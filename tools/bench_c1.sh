#!/bin/bash
# C1 (batch 1, GANet-S TuSimple) latency lines: fp32 and bf16, eager and under a HIP graph, into
# gpurun_out/lines_<tag>/bench_c1_<dtype>_<launch>.json (copy to profiles/).
TAG=$1; shift
OUT=gpurun_out/lines_$TAG
mkdir -p $OUT
for dt in fp32 bf16; do
  for mode in eager graph graphfork; do   # graph: one chain; graphfork: LFA and decode as two branches
    extra="--no-backward --steps 2000 --warmup 200 --spinup 2000 --single-stream"
    [ $mode = graph ] && extra="$extra --graph"
    [ $mode = graphfork ] && extra="$extra --graph --graph-fork"
    f=$OUT/bench_c1_${dt}_${mode}.json
    timeout -k 10 300 python bench.py --workload c1 --dtype $dt $extra "$@" > $f 2> ${f%.json}.err || { echo "FAILED c1 $dt $mode"; tail -5 ${f%.json}.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1])
cb=d.get("cpu_baseline") or {}
print("c1 $dt $mode", round(d["us_per_step"],2), "us/step  lfa", round(d["lfa_ms"]*1e3,2), "us decode", round(d["decode_ms"]*1e3,2), "us  cpu 1 thread", (cb.get("single_thread") or {}).get("value"), "img/s  parity", d.get("parity"))
PY
  done
done
