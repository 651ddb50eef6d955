#!/bin/bash
# PLONK parity + batch-512 bench after a change to plonk.hip
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/plonk
timeout -k 10 500 python -m pytest tests/test_gpu_plonk.py -x -q > gpurun_out/plonk/tests.log 2>&1 && \
timeout -k 10 400 python bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/plonk/bench.json 2> gpurun_out/plonk/bench.err
tail -3 gpurun_out/plonk/tests.log; cat gpurun_out/plonk/bench.json
