# same-box A/B of the four-pass solve (z direction riding on the y passes) against the five-pass one: whole chained RK44 step at 256^3 and 512^3
set -x
O=gpurun_out/r03j
mkdir -p $O
python3 tools/step_lab.py 256 five: four:INS_YZ_FUSED=1 > $O/step_lab_256.txt 2>&1
python3 tools/step_lab.py 512 five: four:INS_YZ_FUSED=1 > $O/step_lab_512.txt 2>&1
grep "^n=" $O/*.txt
