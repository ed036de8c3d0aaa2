#!/bin/bash
# chunk budget of the contract-then-sample pair (GANET_CTS_BUDGET_MB) on the larger workloads
mkdir -p gpurun_out
for wl in c4 c5; do
for mb in 80 112 144 176 224; do
  GANET_CTS_BUDGET_MB=$mb timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-backward --steps 30 --warmup 5 --spinup 40 > gpurun_out/ab_b.json 2> gpurun_out/ab_b.err || { echo FAILED; tail -3 gpurun_out/ab_b.err; exit 1; }
  python - "$wl budget $mb" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "lfa", round(d["lfa_ms"],4))
PY
done; done
