#!/bin/bash
# Host code only (no GPU, no launch): every size / plan query of the library under ASan + UBSan.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/ganet_asan
mkdir -p $OUT
SRC="api decode deform_conv deform_conv_cts deform_conv_stream deform_conv_ring deform_conv_bwd lane_format"
OBJS=""
for s in $SRC; do
  /opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -Iinclude -Iganet_amd/csrc -ffp-contract=off \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer \
    -c ganet_amd/csrc/$s.hip -o $OUT/$s.o &
  OBJS="$OBJS $OUT/$s.o"
done
wait
/opt/rocm/bin/hipcc -O1 -g --offload-arch=gfx950 -std=c++17 -Iinclude \
  -Xarch_host -fsanitize=address,undefined $OBJS tools/plan_fuzz.cc -o $OUT/plan_fuzz
ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 $OUT/plan_fuzz
