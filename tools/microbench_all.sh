#!/bin/bash
# Runs every mode of tools/microbench.py once and writes the record profiles/r<N>_microbench.txt is copied from.
# Usage (on the GPU box): bash tools/microbench_all.sh gpurun_out/r2_microbench.txt
set -e
out=$1
cd "$(dirname "$0")/.."
: > "$out"
for m in gibbs rwm block block_mid shape mid gibbs_mid gibbs_cross gibbs_big user user_wide user_wide_t c1; do
    echo "## tools/microbench.py $m" >> "$out"
    timeout -k 10 280 python3 tools/microbench.py $m >> "$out" 2>&1
done
echo done
