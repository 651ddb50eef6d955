#!/bin/bash
# PLONK batch-512 bench, plain and under rocprofv3 --kernel-trace --stats
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/plonk
timeout -k 10 400 python bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/plonk/bench.json 2> gpurun_out/plonk/bench.err && \
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/plonk/prof -o p -- python3 bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 --cpu-sample 0 > gpurun_out/plonk/bench_rocprof.json 2> gpurun_out/plonk/bench_rocprof.err
cat gpurun_out/plonk/bench.json
