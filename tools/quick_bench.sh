#!/bin/bash
# quick perf probe on the GPU box: C3 (bf16 LFA + decode) and C2 (fp32 LFA)
set -e
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-backward > gpurun_out/bench_c3.json 2> gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c3.json')); print('c3', round(d['value']), 'img/s lfa_ms', round(d['lfa_ms'],4), 'dec_ms', round(d['decode_ms'],4), 'frac', round(d['roofline']['frac'],4))"
timeout -k 10 300 python bench.py --workload c2 --steps 50 --warmup 10 --no-cpu-baseline --no-backward > gpurun_out/bench_c2.json 2>> gpurun_out/bench.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c2.json')); print('c2', round(d['value']), 'img/s lfa_ms', round(d['lfa_ms'],4), 'frac', round(d['roofline']['frac'],4))"
