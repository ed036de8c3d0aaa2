#!/bin/bash
# A/B of the bench step with 2/3/4 result blocks and 2/3 consume threads (interleaved).
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "2 2" "3 2" "4 2" "4 3" "2 2 --no-consume"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-backward --result-blocks $1 --consume-threads $2 $3 > gpurun_out/ab_blocks.json 2> gpurun_out/ab_blocks.err || { echo FAILED; tail -3 gpurun_out/ab_blocks.err; exit 1; }
  python - "$cfg" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_blocks.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "lfa", round(d["lfa_ms"],4), "value", round(d["value"]))
PY
done; done
