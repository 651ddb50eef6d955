#!/bin/bash
# artefacts committed under profiles/: default bench line, rocprofv3 kernel statistics of the same
# command, smoke()
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.log 2>&1
timeout -k 10 500 python3 bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -o bench -- python3 bench.py --cpu-sample 0 --steps 8 --worst-case-steps 0 --bounded-gb 0 --gen-workers 1 > gpurun_out/final/bench_under_rocprof.json 2> gpurun_out/final/rocprof.err
