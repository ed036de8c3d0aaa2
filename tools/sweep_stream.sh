#!/bin/bash
# A/B sweep of the persistent LFA kernel's flow knobs (development tool).
export PYTHONPATH=.
run() { echo "== $*"; env "$@" ONLY3=1 timeout -k 10 60 python -u tools/try_stream.py 128 30 2>&1 | grep -v amdgpu.ids || exit 1; }
for v in ${SWEEP:-0 1 2 3 4 8}; do run ${KNOB:-GANET_STREAM_LEAD_IMAGES}=$v $EXTRA || exit 1; done
