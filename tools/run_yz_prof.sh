set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k
mkdir -p $O
export INS_BENCH_SKIP_K1_512=1 INS_BENCH_SKIP_STRONG_512=1
rocprofv3 --kernel-trace --stats -d $O/s256 -o b --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/s256.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/s512 -o b --output-format csv -- python3 bench.py --n 512 --steps 6 --warmup 2 --no-cpu-baseline > $O/s512.log 2>&1
for f in $O/s256/b_kernel_stats.csv $O/s512/b_kernel_stats.csv; do echo $f; head -12 $f | cut -d, -f1-4 | sed -E 's/\(anonymous namespace\):://g' | cut -c1-150; done
