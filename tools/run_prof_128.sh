set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export INS_BENCH_SKIP_K1_512=1 INS_BENCH_SKIP_STRONG_512=1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof128 -o b --output-format csv -- python3 bench.py --n 128 --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/prof128.log 2>&1
tail -c 600 gpurun_out/prof128.log
