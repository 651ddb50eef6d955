#!/bin/bash
# HBM traffic of the MSM kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md), one non-pipelined step.  Output: gpurun_out/pmc/{fetch,write}/...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$c -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-pipeline > gpurun_out/pmc/$c.json 2> gpurun_out/pmc/$c.err
done
timeout -k 10 400 python3 bench.py --populated 159 --cpu-sample 0 > gpurun_out/pmc/pop159.json 2> gpurun_out/pmc/pop159.err
timeout -k 10 400 python3 bench.py --no-pipeline --steps 6 --cpu-sample 0 > gpurun_out/pmc/nopipe.json 2> gpurun_out/pmc/nopipe.err
