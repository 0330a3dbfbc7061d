set -x
for v in "" "INS_FIELDS_NOBAR=1" "INS_FIELDS_ROWS=2" "INS_FIELDS_ROWS=2 INS_FIELDS_NOBAR=1" "INS_FIELDS_ROWS=3 INS_FIELDS_NOBAR=1" "INS_FIELDS_ROWS=4 INS_FIELDS_NOBAR=1"; do
  echo "== $v"
  env $v timeout -k 10 100 python tools/fields_bench.py 256 | grep -E "^(dissipation_from_strain|eig2field|smagtensor|divoftensor) "
done
