#!/bin/bash
# the round-end GPU suite, with per-test durations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=25 > gpurun_out/suite/tests.log 2>&1
rc=$?
tail -40 gpurun_out/suite/tests.log
exit $rc
