# Round-3 kernel stats of config 5 (cavity 256^3), the all-walls box with the temperature equation and the periodic extended loops (temperature / closure)
set -x
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ext; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/cav -o k --output-format csv -- python3 tools/cavity_prof.py 256 5 > $O/cav.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/wt -o k --output-format csv -- python3 tools/walls_temp_prof.py 256 5 > $O/wt.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/ext_temp -o k --output-format csv -- python3 tools/ext_prof.py 256 temp > $O/ext_temp.log 2>&1 &&
rocprofv3 --kernel-trace --stats -d $O/ext_smag -o k --output-format csv -- python3 tools/ext_prof.py 256 smag > $O/ext_smag.log 2>&1
grep -h "ms/step" $O/*.log
