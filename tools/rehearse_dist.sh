# Rehearsal of `bench.py --gpus 2` on ONE GPU through bench.py's own rank launcher (self_launch): two ranks share the device, exchanges staged through the
# host (gloo).  Checks the driver path (weak line + strong_512 leg on a reduced strong grid) end to end; the numbers mean nothing.
set -x
export INS_BENCH_BACKEND=gloo INS_BENCH_STRONG_GRID=128x128x128 HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline
