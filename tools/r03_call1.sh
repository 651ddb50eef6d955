#!/bin/bash
# round 3, GPU call 1: witness entry tests + bench of the new entry points + launcher on the box
set -o pipefail
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_witness_entry.py -x -q > $O/pytest_witness.log 2>&1 || { tail -30 $O/pytest_witness.log; exit 1; }
tail -3 $O/pytest_witness.log
timeout -k 10 300 python bench.py --steps 10 --cpu-sample 0 --worst-case-steps 0 --verbose > $O/bench_inputs.json 2> $O/bench_inputs.err || { tail -20 $O/bench_inputs.err; exit 1; }
for e in witness witness-abc; do for m in pageable pinned; do
  timeout -k 10 400 python bench.py --steps 10 --cpu-sample 0 --entry $e --host-mem $m --verbose > $O/bench_${e}_${m}.json 2> $O/bench_${e}_${m}.err || { tail -20 $O/bench_${e}_${m}.err; exit 1; }
done; done
timeout -k 10 400 python bench.py --steps 10 --cpu-sample 0 --entry witness-abc --host-mem pageable --copy-threads 12 > $O/bench_witness-abc_pageable_t12.json 2> $O/bench_witness-abc_t12.err || exit 1
timeout -k 10 500 python bench.py --gpus 2 --dist-backend gloo --device 0 --table-budget-gb 60 --batch 512 --steps 6 --cpu-sample 0 --worst-case-steps 0 --verbose > $O/bench_gpus2_gloo.json 2> $O/bench_gpus2_gloo.err || { tail -20 $O/bench_gpus2_gloo.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value'],1), d['n_gpus'], d.get('host_transfer'), d.get('startup_s'))
    except Exception as e: print(f, 'ERR', e)
PY
