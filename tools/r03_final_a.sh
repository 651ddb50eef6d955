#!/bin/bash
# round-3 artefacts, part A: headline line, rocprofv3 kernel statistics of the same command, PMC
# traffic passes, unpipelined line (copied into profiles/ afterwards)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final3
mkdir -p $O gpurun_out/pmc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { cat $O/smoke.log; exit 1; }
timeout -k 10 500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail $O/bench_default.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --cpu-sample 0 --steps 8 --worst-case-steps 0 --bounded-gb 0 --gen-workers 1 > $O/bench_under_rocprof.json 2> $O/rocprof.err || { tail $O/rocprof.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc/$c -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-pipeline --worst-case-steps 0 --bounded-gb 0 --gen-workers 1 > gpurun_out/pmc/$c.json 2> gpurun_out/pmc/$c.err || { tail gpurun_out/pmc/$c.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --no-pipeline --steps 20 --cpu-sample 0 --worst-case-steps 0 > $O/bench_no_pipeline.json 2> $O/bench_no_pipeline.err || exit 1
python3 - <<'PY'
import json
for f in ("bench_default","bench_under_rocprof","bench_no_pipeline"):
    d=json.loads(open(f"gpurun_out/final3/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["value"],1), d.get("value_worst_case"), d.get("bounded"), d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], d["roofline"]["alu"]["frac"])
PY
head -8 $O/bench_kernel_stats.csv | cut -c1-160
