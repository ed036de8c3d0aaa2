#!/bin/bash
# A/B sweep of the persistent LFA kernel's flow knobs (development tool).
export PYTHONPATH=.
run() { echo "== $*"; env "$@" ONLY3=1 timeout -k 10 60 python -u tools/try_stream.py 128 30 2>&1 | grep -v amdgpu.ids || exit 1; }
run GANET_STREAM_DEBUG_BITS=0 || exit 1
run GANET_STREAM_DEBUG_BITS=1 || exit 1
run GANET_STREAM_DEBUG_BITS=2 || exit 1
run GANET_STREAM_DEBUG_BITS=3 || exit 1
run GANET_STREAM_DEBUG_BITS=5 || exit 1
run GANET_STREAM_DEBUG_BITS=6 || exit 1
run GANET_STREAM_DEBUG_BITS=7 || exit 1
