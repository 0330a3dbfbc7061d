#!/bin/bash
# A/B of the 64-outputs-per-wavefront K1 (ins_flux64.hip) against the 62-output kernel, plus its row / z-chunk knobs.
cd "$(dirname "$0")/.."
B='import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])'
echo "== K1 alone (momentum_), old kernel"
INS_DISABLE_FLUX64=1 python tools/k1_time.py 256 512
for lds in ${LDSS:-0 70000}; do for r in ${ROWS:-4 5 6}; do for zc in ${ZCS:-8 16}; do
  echo "== K1 alone flux64 rows=$r zc=$zc lds=$lds"
  INS_FLUX64_LDS=$lds INS_FLUX64_ROWS=$r INS_FLUX64_ZC=$zc python tools/k1_time.py 256 512
  echo "== skeleton rows=$r zc=$zc lds=$lds"
  INS_FLUX64_LDS=$lds INS_FLUX64_SKEL=1 INS_FLUX64_ROWS=$r INS_FLUX64_ZC=$zc python tools/k1_time.py 256 512
done; done; done
echo "== bench 256 old"
INS_DISABLE_FLUX64=1 INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
for rc in ${ROWSC:-2 3 4}; do for zc in ${ZCS:-8 16}; do for r in 4 6; do
  echo "== bench 256 flux64 rows=$r rows_corr=$rc zc=$zc"
  INS_FLUX64_ROWS=$r INS_FLUX64_ROWS_CORR=$rc INS_FLUX64_ZC=$zc INS_BENCH_SKIP_K1_512=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline | python -c "$B"
done; done; done
