#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/emul
timeout -k 10 900 python -m pytest tests/test_gpu_emulated.py -x -q --durations=8 -k malformed > gpurun_out/emul/tests.log 2>&1 || { tail -40 gpurun_out/emul/tests.log; exit 1; }
tail -5 gpurun_out/emul/tests.log

