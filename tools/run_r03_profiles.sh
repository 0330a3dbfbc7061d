# Round-3 profile set (tracked copies go to profiles/r03_*): kernel stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE separately, MI355X_MICROARCH.md) of the
# bench command at 256^3 and of the SAME program on the strong-scaling box (--n 512 = BASELINE configs[3] on one GPU).  The program goes directly behind `--`.
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03p
mkdir -p $O
export INS_BENCH_SKIP_K1_512=1 INS_BENCH_SKIP_STRONG_512=1
rocprofv3 --kernel-trace --stats -d $O/s512 -o b --output-format csv -- python3 bench.py --n 512 --steps 10 --warmup 2 --no-cpu-baseline > $O/s512.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f512 -o f --output-format csv -- python3 bench.py --n 512 --steps 4 --warmup 1 --no-cpu-baseline > $O/f512.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w512 -o w --output-format csv -- python3 bench.py --n 512 --steps 4 --warmup 1 --no-cpu-baseline > $O/w512.log 2>&1 &&
python3 tools/pmc_traffic.py $O/f512/f_counter_collection.csv $O/w512/w_counter_collection.csv $O/r03_pmc_traffic_512.json "python3 bench.py --n 512 --steps 4 --warmup 1 --no-cpu-baseline" "TGV3D 512^3" > $O/pmc_table_512.txt 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/s256 -o b --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/s256.log 2>&1 &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f256 -o f --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/f256.log 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w256 -o w --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/w256.log 2>&1 &&
python3 tools/pmc_traffic.py $O/f256/f_counter_collection.csv $O/w256/w_counter_collection.csv $O/r03_pmc_traffic.json "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline" "TGV3D 256^3" > $O/pmc_table_256.txt 2>&1
cat $O/pmc_table_512.txt $O/pmc_table_256.txt
find $O -name "*kernel_stats.csv" | head
unset INS_BENCH_SKIP_K1_512 INS_BENCH_SKIP_STRONG_512
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
