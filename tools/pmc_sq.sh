#!/bin/bash
# Issue / wait / clock view of the MSM kernels: SQ and GRBM counters in one rocprofv3 --pmc pass
# over one non-pipelined step (effective clock = GRBM_GUI_ACTIVE / 8 / kernel wall time,
# MI355X_MICROARCH.md "DVFS give-back").  Output: gpurun_out/pmc_sq/
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_sq
timeout -k 10 500 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_sq/run -o pmc -- python3 bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-pipeline --worst-case-steps 0 --bounded-gb 0 --gen-workers 1 "$@" > gpurun_out/pmc_sq/bench.json 2> gpurun_out/pmc_sq/bench.err
