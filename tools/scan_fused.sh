for rows in 2 3 4; do for zc in 4 8 16; do for xw in 2 4; do
  r=$(INS_FLUX_ROWS=$rows INS_FLUX_ZC=$zc INS_FLUX_XW=$xw python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))")
  echo "rows=$rows zc=$zc xw=$xw : $r"
done; done; done
