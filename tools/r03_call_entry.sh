#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/entry
timeout -k 10 900 python -m pytest tests/test_gpu_witness_entry.py tests/test_gpu_commitment.py::test_two_commitments -x -q > gpurun_out/entry/tests.log 2>&1 || { tail -40 gpurun_out/entry/tests.log; exit 1; }
tail -3 gpurun_out/entry/tests.log
timeout -k 10 600 python bench.py --entry witness --cpu-sample 0 --worst-case-steps 0 --bounded-gb 0 > gpurun_out/entry/bench_witness_pageable.json 2> gpurun_out/entry/bench_witness.err
timeout -k 10 600 python bench.py --entry witness --host-mem pinned --cpu-sample 0 --worst-case-steps 0 --bounded-gb 0 > gpurun_out/entry/bench_witness_pinned.json 2> gpurun_out/entry/bench_witness_pinned.err
timeout -k 10 600 python bench.py --cpu-sample 0 --worst-case-steps 0 --bounded-gb 0 > gpurun_out/entry/bench_inputs.json 2> gpurun_out/entry/bench_inputs.err
python - <<'PY'
import json
for n in ('witness_pageable','witness_pinned','inputs'):
    d=json.load(open(f'gpurun_out/entry/bench_{n}.json')); print(n, round(d['value'],1), round(d['ms_per_step'],1), d.get('host_transfer'))
PY
