#!/bin/bash
# delta multiples on the assembly stream: prove tests, then headline with and without the old library?  (A/B is
# across builds, so only the new one is measured here, on 20 + 20 steps)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dstream
timeout -k 10 900 python -m pytest tests/test_gpu_prove.py tests/test_gpu_benched.py tests/test_gpu_commitment.py::test_two_commitments tests/test_gpu_witness_entry.py -x -q > gpurun_out/dstream/tests.log 2>&1 || { tail -40 gpurun_out/dstream/tests.log; exit 1; }
tail -3 gpurun_out/dstream/tests.log
for i in 1 2; do
timeout -k 10 400 python3 bench.py --steps 20 --cpu-sample 0 --worst-case-steps 0 --bounded-gb 0 > gpurun_out/dstream/b$i.json 2> gpurun_out/dstream/b$i.err
python3 - <<PY
import json
d=json.load(open("gpurun_out/dstream/b$i.json")); print(round(d["value"],1), round(d["ms_per_step"],1), {k: round(v,1) for k,v in d["stage_ms"].items()})
PY
done
