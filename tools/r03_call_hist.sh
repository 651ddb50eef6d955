#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/hist
timeout -k 10 900 python -m pytest tests/test_gpu_commitment.py tests/test_gpu_emulated.py -x -q > gpurun_out/hist/tests.log 2>&1 || { tail -40 gpurun_out/hist/tests.log; exit 1; }
tail -3 gpurun_out/hist/tests.log
timeout -k 10 700 python bench.py --workload emulated-poseidon --batch 512 --steps 3 --warmup 1 --cpu-sample 2 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/hist/bench_emul.json 2> gpurun_out/hist/bench_emul.err
timeout -k 10 700 python bench.py --workload address-bytes --batch 512 --steps 3 --warmup 1 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/hist/bench_bytes.json 2> gpurun_out/hist/bench_bytes.err
timeout -k 10 700 python bench.py --workload address-commit --steps 5 --warmup 2 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/hist/bench_commit.json 2> gpurun_out/hist/bench_commit.err
python - <<'PY'
import json
for n in ('emul','bytes','commit'):
    d=json.load(open(f'gpurun_out/hist/bench_{n}.json')); print(n, round(d['value'],1), d['ms_per_step'], d['stage_ms'])
PY
