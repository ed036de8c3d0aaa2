#!/bin/bash
# A/B of the bench step with HIP events around every / every 4th / every 16th step (interleaved).
mkdir -p gpurun_out
for rep in 1 2 3; do
for cfg in "1" "4" "16" "1 --no-consume" "16 --no-consume"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-backward --event-every $1 $2 > gpurun_out/ab_ev.json 2> gpurun_out/ab_ev.err || { echo FAILED; tail -3 gpurun_out/ab_ev.err; exit 1; }
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_ev.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "lfa", round(d["lfa_ms"],4), "dec", round(d["decode_ms"],4), "value", round(d["value"]))
PY
done; done
