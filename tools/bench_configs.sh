#!/bin/bash
# secondary configurations of BASELINE.json through bench.py --workload
set -e
out=gpurun_out/configs
mkdir -p $out
run() { name=$1; shift; timeout -k 10 500 python bench.py --cpu-sample 0 --steps 4 --warmup 1 --worst-case-steps 0 "$@" > $out/$name.json 2> $out/$name.err; echo "$name done"; }
run poseidon --workload poseidon --batch 8192 --distinct 256
run elgamal-add --workload elgamal-add --batch 8192 --distinct 128
run elgamal-encrypt --workload elgamal-encrypt --batch 4096 --distinct 64
run verifier --workload verifier --batch 1024
run address --workload address --batch 1024 --distinct 32
# PLONK backend (config 5 as BASELINE.json words it, and the small circuits)
run plonk-poseidon --backend plonk --workload poseidon --batch 1024 --distinct 64
run plonk-address --backend plonk --workload address --batch 512 --distinct 8 --steps 2
