#!/bin/bash
set -e
out=gpurun_out/ilp
mkdir -p $out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1
timeout -k 10 300 python bench.py --cpu-sample 0 > $out/arbo.json 2> $out/arbo.err
timeout -k 10 300 python bench.py --cpu-sample 0 --no-pipeline --steps 4 > $out/arbo_nopipe.json 2> $out/arbo_nopipe.err
timeout -k 10 400 python bench.py --workload address --batch 512 --distinct 32 --cpu-sample 0 --steps 4 --warmup 1 > $out/address512.json 2> $out/address512.err
timeout -k 10 400 python bench.py --workload elgamal-encrypt --batch 4096 --distinct 64 --cpu-sample 0 --steps 4 --warmup 1 > $out/encrypt.json 2> $out/encrypt.err
