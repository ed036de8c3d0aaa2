#!/bin/bash
# A/B of one environment knob on the default bench (interleaved): tools/ab_env.sh NAME=VALUE [bench args]
KV=$1; shift
mkdir -p gpurun_out
for rep in 1 2 3; do
  for on in 0 1; do
    if [ $on = 1 ]; then export "$KV"; else unset "${KV%%=*}"; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-backward "$@" > gpurun_out/ab_env.json 2> gpurun_out/ab_env.err || { echo FAILED; tail -3 gpurun_out/ab_env.err; exit 1; }
    python - "$on $KV" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_env.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "lfa", round(d["lfa_ms"],4), "value", round(d["value"]))
PY
  done
done
