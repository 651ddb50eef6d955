#!/bin/bash
set -o pipefail
O=gpurun_out/r3d
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --cpu-sample 0 --steps 20 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
for w in address address-commit; do
  timeout -k 10 400 python bench.py --workload $w --distinct 64 --steps 6 --cpu-sample 8 --verbose > $O/bench_$w.json 2> $O/bench_$w.err || { tail -20 $O/bench_$w.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3d/*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],1), d.get('bounded'), {k:round(v,1) for k,v in d['stage_ms'].items()}, d.get('cpu_baseline'))
PY
