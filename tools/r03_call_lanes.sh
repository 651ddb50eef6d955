#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
timeout -k 10 900 python -m pytest tests/test_gpu_parity_holes.py::test_solver_every_lane_count_vs_oracle tests/test_gpu_commitment.py::test_range_circuit tests/test_gpu_emulated.py -x -q > gpurun_out/lanes/tests.log 2>&1 || { tail -40 gpurun_out/lanes/tests.log; exit 1; }
tail -3 gpurun_out/lanes/tests.log
for w in emulated-poseidon address-bytes; do
timeout -k 10 700 python bench.py --workload $w --batch 512 --steps 3 --warmup 1 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/lanes/bench_$w.json 2> gpurun_out/lanes/bench_$w.err
done
for w in address address-commit; do
timeout -k 10 700 python bench.py --workload $w --steps 5 --warmup 2 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/lanes/bench_$w.json 2> gpurun_out/lanes/bench_$w.err
done
python - <<'PY'
import json
for n in ('emulated-poseidon','address-bytes','address','address-commit'):
    d=json.load(open(f'gpurun_out/lanes/bench_{n}.json')); print(n, round(d['value'],1), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['stage_ms'].items()})
PY
grep -h "lanes" gpurun_out/lanes/*.err | head
