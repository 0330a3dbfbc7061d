set -x
O=gpurun_out/r03l
mkdir -p $O
python3 tools/step_lab.py 256 five: four:INS_YZ_FUSED=1 four_nofft:INS_YZ_FUSED=1,INS_YZ_SKEL=1 four_norec:INS_YZ_FUSED=1,INS_YZ_SKEL=2 > $O/step_lab_256.txt 2>&1
python3 tools/step_lab.py 512 five: four:INS_YZ_FUSED=1 four_nofft:INS_YZ_FUSED=1,INS_YZ_SKEL=1 four_norec:INS_YZ_FUSED=1,INS_YZ_SKEL=2 > $O/step_lab_512.txt 2>&1
grep "^n=" $O/*.txt
