#!/bin/bash
set -e
out=gpurun_out/configs
mkdir -p $out
run() { name=$1; shift; timeout -k 10 400 python bench.py --workload $name --cpu-sample 0 --steps 4 --warmup 1 "$@" > $out/$name.json 2> $out/$name.err; echo "$name done"; }
run poseidon --batch 8192 --distinct 256
run elgamal-add --batch 8192 --distinct 128
run elgamal-encrypt --batch 4096 --distinct 64
