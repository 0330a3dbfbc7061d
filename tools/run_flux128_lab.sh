# same-box A/B of the two-columns-per-lane stage kernels (csrc/ins_flux128.hip) against the one-column ones: plain K1 and the whole chained RK44 step
set -x
O=gpurun_out/r03e
mkdir -p $O
python3 tools/k1_lab.py 512 one:INS_DISABLE_FLUX128=1 two: two_nw4:INS_FLUX128_NW=4 two_r1:INS_FLUX128_ROWS=1 two_xw4:INS_FLUX128_XW=4 two_xw1:INS_FLUX128_XW=1 two_zc32:INS_FLUX128_ZC=32 > $O/k1_lab_512.txt 2>&1
python3 tools/k1_lab.py 256 one:INS_DISABLE_FLUX128=1 two: two_nw8:INS_FLUX128_NW=8 two_r1:INS_FLUX128_ROWS=1 two_zc32:INS_FLUX128_ZC=32 two_zc16:INS_FLUX128_ZC=16 > $O/k1_lab_256.txt 2>&1
python3 tools/step_lab.py 256 one:INS_DISABLE_FLUX128=1 two: two_cr1:INS_FLUX128_ROWS_CORR=1 two_nw4:INS_FLUX128_NW=4 two_cr1_nw4:INS_FLUX128_ROWS_CORR=1,INS_FLUX128_NW=4 two_zc32:INS_FLUX128_ZC=32 two_cr1_zc32:INS_FLUX128_ROWS_CORR=1,INS_FLUX128_ZC=32 > $O/step_lab_256.txt 2>&1
python3 tools/step_lab.py 512 one:INS_DISABLE_FLUX128=1 two: two_cr1:INS_FLUX128_ROWS_CORR=1 two_cr1_xw4:INS_FLUX128_ROWS_CORR=1,INS_FLUX128_XW=4 two_xw4:INS_FLUX128_XW=4 > $O/step_lab_512.txt 2>&1
tail -n 12 $O/*.txt
