#!/bin/bash
# round-3 artefacts, part B: the other BASELINE configs, PLONK kernel statistics, one-rank RCCL strong
# scaling runs, two ranks on one device through the self-launcher
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final3b
mkdir -p $O
run() { name=$1; shift; timeout -k 10 500 python3 bench.py --cpu-sample 0 --steps 4 --warmup 1 --worst-case-steps 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; return 1; }; echo "$name done"; }
run poseidon --workload poseidon --batch 8192 --distinct 256 || exit 1
run elgamal-add --workload elgamal-add --batch 8192 --distinct 32 || exit 1
run elgamal-encrypt --workload elgamal-encrypt --batch 4096 --distinct 16 || exit 1
run verifier --workload verifier --batch 1024 || exit 1
run address --workload address --batch 1024 --distinct 32 || exit 1
run address-commit --workload address-commit --batch 1024 --distinct 32 || exit 1
run plonk-poseidon --backend plonk --workload poseidon --batch 1024 --distinct 64 || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o plonk -- python3 bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 > $O/plonk_512_under_rocprof.json 2> $O/plonk_512.err || { tail -20 $O/plonk_512.err; exit 1; }
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/plonk_512_kernel_stats.csv
timeout -k 10 400 python3 bench.py --backend plonk --workload address --batch 512 --steps 3 --warmup 1 > $O/plonk-address.json 2> $O/plonk-address.err || exit 1
timeout -k 10 500 python3 bench.py --scaling strong --workload verifier --batch 4096 --force-collective --steps 4 --cpu-sample 0 --worst-case-steps 0 > $O/strong_verifier4096_1rank.json 2> $O/strong_verifier.err || { tail $O/strong_verifier.err; exit 1; }
timeout -k 10 500 python3 bench.py --gpus 2 --dist-backend gloo --device 0 --table-budget-gb 60 --batch 512 --steps 6 --cpu-sample 0 --worst-case-steps 0 > $O/gpus2_gloo_one_device.json 2> $O/gpus2.err || { tail $O/gpus2.err; exit 1; }
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final3b/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['value'],1), d['n_gpus'], d.get('gathered_on_every_rank'))
    except Exception as e: print(f, 'ERR', e)
PY
