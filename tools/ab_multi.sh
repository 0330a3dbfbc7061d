#!/bin/bash
# A/B several builds of libinship.so in one gpurun call (same box), interleaved: tools/ab_multi.sh lib1.so lib2.so ...   ("-" = the in-tree build)
cd "$(dirname "$0")/.."
B='import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline_k1"]["avg_launch_ms"])'
for i in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "
    if [ "$lib" = "-" ]; then INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
    else INS_HIP_LIB=$PWD/$lib INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"; fi
  done
done
