#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_plonk.py -x -q --durations=10 > $O/pytest_plonk.log 2>&1 || { tail -40 $O/pytest_plonk.log; exit 1; }
tail -15 $O/pytest_plonk.log
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o plonk -- python3 bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 --verbose > $O/plonk_512_under_rocprof.json 2> $O/plonk_512.err || { tail -20 $O/plonk_512.err; exit 1; }
cat $O/plonk_512_under_rocprof.json | head -c 600
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/plonk_512_kernel_stats.csv
head -25 $O/plonk_512_kernel_stats.csv | cut -c1-200
