# Round-2 profile set of the bench command: kernel stats + the two PMC passes (FETCH_SIZE, WRITE_SIZE separately, MI355X_MICROARCH.md)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export INS_BENCH_SKIP_K1_512=1 INS_BENCH_SKIP_STRONG_512=1
rocprofv3 --kernel-trace --stats -d gpurun_out/pb_stats -o b --output-format csv -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/pb_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pb_f -o f --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pb_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pb_w -o w --output-format csv -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pb_w.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pb_f/f_counter_collection.csv gpurun_out/pb_w/w_counter_collection.csv gpurun_out/r02h_pmc_traffic.json "python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline" "TGV3D 256^3" > gpurun_out/pmc_table.txt 2>&1
cat gpurun_out/pmc_table.txt
unset INS_BENCH_SKIP_K1_512 INS_BENCH_SKIP_STRONG_512
python3 bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; cat gpurun_out/bench3.json
