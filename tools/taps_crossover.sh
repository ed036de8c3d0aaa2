#!/bin/bash
# Where algorithm 5 (taps on different wavefronts) stops paying: LFA time under a HIP graph at
# the C1 shape for growing batches, automatic choice forced off (GANET_TAPS_MAX_TILES=0) and on
# (=100000).  Output: gpurun_out/taps_crossover.txt
OUT=gpurun_out/taps_crossover.txt
: > $OUT
for dt in bf16 fp32; do
  for b in ${BATCHES:-1 2 3 4 6 8 12 16}; do
    for lim in 0 100000; do
      GANET_TAPS_MAX_TILES=$lim timeout -k 10 120 python bench.py --workload c1 --dtype $dt --batch $b --no-backward \
        --no-cpu-baseline --steps 1000 --warmup 100 --spinup 500 --single-stream --graph > /tmp/x.json 2> /tmp/x.err || { tail -3 /tmp/x.err; exit 1; }
      python - >> $OUT <<PY
import json
d=json.loads(open("/tmp/x.json").read().strip().splitlines()[-1])
print("$dt B=$b limit=$lim algo", d["roofline"]["lfa_algorithm"], "lfa_us %.2f step_us %.2f" % (d["lfa_ms"]*1e3, d["us_per_step"]))
PY
    done
  done
done
cat $OUT
