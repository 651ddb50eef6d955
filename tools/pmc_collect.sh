#!/bin/bash
# HBM traffic of the MSM kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md), one non-pipelined step.  Output: gpurun_out/pmc/{fetch,write}/...
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$c -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-pipeline --worst-case-steps 0 --bounded-gb 0 --gen-workers 1 > gpurun_out/pmc/$c.json 2> gpurun_out/pmc/$c.err
done
