#!/bin/bash
# ablation of the fused LFA kernel (profiling only): full / no MFMA / no loads / neither / zero offsets
for wl in c3 c2; do
  for ab in 0 1 2 3; do
    python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-backward --ablate $ab 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl ablate=$ab lfa_ms', round(d['lfa_ms'],4))"
  done
  python bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline --no-backward --pts zero 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$wl pts=zero lfa_ms', round(d['lfa_ms'],4))"
done
