set -x
O=gpurun_out/r03f
mkdir -p $O
python3 tools/step_lab.py 256 one:INS_DISABLE_FLUX128=1 two_occ1: two_cr1:INS_FLUX128_ROWS_CORR=1 two_occ1_zc32:INS_FLUX128_ZC=32 two_occ1_zc16:INS_FLUX128_ZC=16 > $O/step_lab_256.txt 2>&1
python3 tools/step_lab.py 512 one:INS_DISABLE_FLUX128=1 two_occ1: two_cr1:INS_FLUX128_ROWS_CORR=1 two_occ1_zc32:INS_FLUX128_ZC=32 two_occ1_xw4:INS_FLUX128_XW=4 > $O/step_lab_512.txt 2>&1
tail -n 8 $O/*.txt
