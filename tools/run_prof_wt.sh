set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/profwt -o w --output-format csv -- python3 tools/walls_temp_prof.py 256 5 > gpurun_out/profwt.log 2>&1
tail -n 1 gpurun_out/profwt.log
