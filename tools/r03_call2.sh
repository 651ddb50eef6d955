#!/bin/bash
set -o pipefail
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_gpu_parity_holes.py -x -q > $O/pytest_holes.log 2>&1 || { tail -40 $O/pytest_holes.log; exit 1; }
tail -3 $O/pytest_holes.log
for g in 32 64 128; do
  timeout -k 10 300 python bench.py --steps 12 --cpu-sample 0 --worst-case-steps 0 --table-budget-gb $g > $O/bench_budget_$g.json 2> $O/bench_budget_$g.err || { tail -20 $O/bench_budget_$g.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); t=d['config']['msm_window_tables']
    print(f.split('/')[-1], round(d['value'],1), t['g1_comb_k'], t['g2_comb_k'], (t['g1_table_bytes']+t['g2_table_bytes'])/1e9)
PY
