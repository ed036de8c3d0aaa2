#!/bin/bash
# Ablation rows of algorithm 4 (csrc/deform_conv_ring.hip) at C3: one variant library per row
# (-DGANET_RING_ABLATE=<bits>: 1 nothing produced, 2 nothing sampled, 4 no waiting at the team
# barrier, 8 no weight staging), built HERE on the CPU by `tools/ring_ablate.sh build`, timed on
# the GPU box by `tools/ring_ablate.sh run [B] [iters]`.
set -e
ROWS="3 11 15 1 2 4 8 9 10"
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  for b in $ROWS; do GANET_RING_DEFS="-DGANET_RING_ABLATE=$b" python -m ganet_amd.build | tail -1; done
  exit 0
fi
B=${2:-128}; IT=${3:-50}
export PYTHONPATH=.
echo "== product library"; timeout -k 10 120 python -u tools/try_ring.py $B $IT
for b in $ROWS; do
  lib=$(GANET_RING_DEFS="-DGANET_RING_ABLATE=$b" python -c "from ganet_amd import build; print(build.build_lib())")
  echo "== ablate bits $b"
  GANET_HIP_LIB=$lib ONLY4=1 timeout -k 10 120 python -u tools/try_ring.py $B $IT
done
