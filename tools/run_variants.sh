#!/bin/bash
# on the GPU box: bench every variants/lib_*.so named (or all): tools/run_variants.sh [bench args] -- name...
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for n in "$@"; do
  export GANET_HIP_LIB=$PWD/variants/lib_$n.so
  timeout -k 10 200 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-backward "${ARGS[@]}" 2>gpurun_out/var_$n.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'step_ms', round(d['ms_per_step'],4), 'lfa_ms', round(d['lfa_ms'],4), 'parity', d.get('parity'))" || { echo "$n FAILED"; tail -3 gpurun_out/var_$n.err; exit 1; }
done
