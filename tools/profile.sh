#!/bin/bash
# Profiles one bench.py workload on the GPU box: rocprofv3 kernel stats, then PMC passes (each counter group
# in a run of its own, as the pool requires).  Usage: tools/profile.sh <tag> [bench.py args...]
# Writes gpurun_out/<tag>_*; tools/summarize_profile.py turns them into profiles/<tag>_*.csv.
set -e
tag=$1; shift
out=gpurun_out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
args="--steps 1 --warmup 0 --no-cpu-baseline --no-ess $*"
timeout -k 10 300 rocprofv3 --output-format csv --kernel-trace --stats -d $out/${tag}_stats -o s -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ess "$@" > $out/${tag}_stats.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/${tag}_pmc_fetch -o p -- python3 bench.py $args > $out/${tag}_pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/${tag}_pmc_write -o p -- python3 bench.py $args > $out/${tag}_pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $out/${tag}_pmc_sq -o p -- python3 bench.py $args > $out/${tag}_pmc_sq.log 2>&1
timeout -s KILL 200 rocprofv3 --output-format csv --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $out/${tag}_pmc_sq2 -o p -- python3 bench.py $args > $out/${tag}_pmc_sq2.log 2>&1
echo done $tag
