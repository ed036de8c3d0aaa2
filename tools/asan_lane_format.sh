#!/bin/bash
# Host code only (no GPU): csrc/lane_format.hip + tools/lane_format_fuzz.cc under ASan + UBSan.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/ganet_asan
mkdir -p $OUT
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -Iinclude -Iganet_amd/csrc -ffp-contract=off \
  -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
  ganet_amd/csrc/lane_format.hip ganet_amd/csrc/api.hip tools/lane_format_fuzz.cc -o $OUT/lane_format_fuzz
ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $OUT/lane_format_fuzz
