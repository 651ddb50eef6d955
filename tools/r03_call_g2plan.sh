#!/bin/bash
# after the planner's G2 sign-pattern price changed: the benched-plan tests, then the headline artefacts
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/g2plan
timeout -k 10 900 python -m pytest tests/test_gpu_benched.py tests/test_gpu_parity_holes.py tests/test_gpu_witness_entry.py -x -q > gpurun_out/g2plan/tests.log 2>&1 || { tail -40 gpurun_out/g2plan/tests.log; exit 1; }
tail -3 gpurun_out/g2plan/tests.log
bash tools/r03_final_a.sh
