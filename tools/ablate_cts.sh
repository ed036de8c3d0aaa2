#!/bin/bash
# ablation of the contract-then-sample LFA kernels (GANET_ABLATE build, profiling only):
# bit0 no MFMA, bit2 Y stores of the first slice only, bit3 no Y gather
for ab in ${ABLS:-0 1 4 5 8 12 13}; do
  python bench.py --workload c3 --steps 30 --warmup 5 --no-cpu-baseline --no-backward --ablate $ab 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 ablate=$ab lfa_ms', round(d['lfa_ms'],4))" || exit 1
done
