set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof512 $R/gpurun_out/prof256
cd $R
python -m pytest tests/test_gpu_bench_shape.py tests/test_gpu_slab.py tests/test_gpu_nccl.py -x -q -m gpu > gpurun_out/t3.log 2>&1; echo "rc=$?" >> gpurun_out/t3.log; tail -3 gpurun_out/t3.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flux64 or chained or full_size" > gpurun_out/t4.log 2>&1; echo "rc=$?" >> gpurun_out/t4.log; tail -3 gpurun_out/t4.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof512 -o p512 --output-format csv -- python3 tools/prof_step.py 512 4 > gpurun_out/prof512.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof256 -o p256 --output-format csv -- python3 tools/prof_step.py 256 10 > gpurun_out/prof256.log 2>&1
ls -R gpurun_out/prof512 | head -20
