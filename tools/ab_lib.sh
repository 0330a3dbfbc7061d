#!/bin/bash
# A/B two builds of libinship.so in one gpurun call (same box): tools/ab_lib.sh build_ab/libinship_prev.so
cd "$(dirname "$0")/.."
B='import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_k1"]["avg_launch_ms"])'
for i in 1 2 3; do
  echo -n "prev: "; INS_HIP_LIB=$PWD/$1 INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
  echo -n "new : "; INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
done
