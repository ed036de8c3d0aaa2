#!/bin/bash
# A fuzz campaign on the box: tools/fuzz_campaign.sh <first seed> <seeds> [forward backward decode lfa]
S0=$1; N=$2; F=${3:-1500}; B=${4:-500}; D=${5:-1500}; L=${6:-120}
mkdir -p gpurun_out/fuzz
for ((s=S0; s<S0+N; s++)); do
  PYTHONPATH=. timeout -k 10 1000 python -u tools/fuzz.py --seed $s --forward $F --backward $B --decode $D --lfa $L > gpurun_out/fuzz/seed_$s.log 2>&1
  rc=$?
  tail -1 gpurun_out/fuzz/seed_$s.log
  grep -c "^FAIL" gpurun_out/fuzz/seed_$s.log | sed "s/^/  FAIL lines: /"
  [ $rc = 0 ] || { grep "^FAIL" gpurun_out/fuzz/seed_$s.log | head -5; exit 1; }
done
