#!/bin/bash
# Phase times of the persistent LFA kernel (library built with GANET_STREAM_DEBUG=1).
export PYTHONPATH=. ONLY3=1 STAMPS=1
for bits in ${BITS:-0}; do
  echo "== debug bits $bits $EXTRA"
  env GANET_STREAM_DEBUG_BITS=$bits $EXTRA timeout -k 10 60 python -u tools/try_stream.py 128 20 2>&1 | grep -v amdgpu.ids || exit 1
done
