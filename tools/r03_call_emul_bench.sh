#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/emul
timeout -k 10 700 python bench.py --workload emulated-poseidon --batch 512 --steps 3 --warmup 1 --cpu-sample 2 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/emul/bench.json 2> gpurun_out/emul/bench.err || { tail -30 gpurun_out/emul/bench.err; exit 1; }
cat gpurun_out/emul/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/emul/prof -o p -- python3 bench.py --workload emulated-poseidon --batch 512 --steps 2 --warmup 1 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 > gpurun_out/emul/bench_rocprof.json 2> gpurun_out/emul/bench_rocprof.err
