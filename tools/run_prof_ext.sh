set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for mode in temp smag; do
rocprofv3 --kernel-trace --stats -d gpurun_out/profext_$mode -o ext --output-format csv -- python3 tools/ext_prof.py 256 $mode 5 > gpurun_out/profext_$mode.log 2>&1
tail -1 gpurun_out/profext_$mode.log
done
