#!/bin/bash
# Tuning variants of algorithm 4 (csrc/deform_conv_ring.hip): each row is a set of -D flags
# for that file; `build` makes one library per row HERE, `run [B] [iters]` times them on the
# GPU box (tools/try_ring.py: time, status word, bit-identity with algorithm 2).
set -e
cd "$(dirname "$0")/.."
ROWS=(
  "-DGANET_RING_ABLATE=27"
  "-DGANET_RING_ABLATE=19"
  "-DGANET_RING_ABLATE=3"
  "-DGANET_RING_ABLATE=1"
  "-DGANET_RING_ABLATE=2"
  "-DGANET_RING_ABLATE=4"
  "-DGANET_RING_GATHER_AUX=0"
)
if [ "$1" = build ]; then
  for r in "${ROWS[@]}"; do GANET_RING_DEFS="$r" python -m ganet_amd.build | tail -1; done
  exit 0
fi
B=${2:-128}; IT=${3:-50}
export PYTHONPATH=.
echo "== product library"; timeout -k 10 120 python -u tools/try_ring.py $B $IT || true
for r in "${ROWS[@]}"; do
  lib=$(GANET_RING_DEFS="$r" python -c "from ganet_amd import build; print(build.build_lib())")
  echo "== $r"
  GANET_HIP_LIB=$lib timeout -k 10 120 python -u tools/try_ring.py $B $IT 2>&1 | grep -v amdgpu.ids | cut -c1-110 || true
done
