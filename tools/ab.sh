#!/bin/bash
# A/B on the GPU box: tools/ab.sh ARM ARM ... -- [bench args]
# an ARM is a quoted group of env settings ("A=1 B=2") and/or bench flags ("--side-decode");
# "base" = nothing.  One discarded warm-up run, then the arms interleaved (rep outer, arm
# inner): the first process on a fresh box runs several % slow.
ARMS=(); PRE=${AB_PRE:-}; while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARMS+=("$1"); shift; done; shift
timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-backward "$@" >/dev/null 2>&1
for rep in 1 2 3; do
for arm in "${ARMS[@]}"; do
  envs=(); flags=()
  for w in $arm; do case "$w" in base) ;; --*) flags+=("$w");; *=*) envs+=("$w");; *) flags+=("$w");; esac; done
  env "${envs[@]}" timeout -k 10 200 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-backward "${flags[@]}" "$@" 2>gpurun_out/ab.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('[$arm]', 'step_ms', round(d['ms_per_step'],4), 'lfa_ms', round(d['lfa_ms'],4), 'dec_ms', round(d['decode_ms'],4))" || { echo "[$arm] FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
done
done | sort
