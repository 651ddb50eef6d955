#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/lanes
for l in 16 32 64; do
timeout -k 10 700 python bench.py --workload emulated-poseidon --batch 512 --steps 3 --warmup 1 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 --solver-lanes $l > gpurun_out/lanes/emul_$l.json 2> gpurun_out/lanes/emul_$l.err
done
for l in 16 32; do
timeout -k 10 700 python bench.py --workload address --steps 5 --warmup 2 --cpu-sample 0 --bounded-gb 0 --worst-case-steps 0 --solver-lanes $l > gpurun_out/lanes/address_$l.json 2> gpurun_out/lanes/address_$l.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/lanes/emul_*.json')+glob.glob('gpurun_out/lanes/address_*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), round(d['ms_per_step'],1), {k: round(v,1) for k,v in d['stage_ms'].items() if k in ('solve','quotient_ntt','main_stream_span')})
PY
