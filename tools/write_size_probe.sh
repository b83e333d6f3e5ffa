#!/bin/bash
# WRITE_SIZE of a small C4-shaped run (8 192 chains x 1 000 iterations, thin 10: 655 MB of trace), for experiments with the
# trace stores of rwm_block.hip.  Usage (GPU box): bash tools/write_size_probe.sh <tag>
set -e
tag=$1
chains=${2:-8192}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --output-format csv --pmc WRITE_SIZE -d gpurun_out/ws_${tag} -o p -- python3 bench.py --workload c4 --chains $chains --iters 1000 --steps 1 --warmup 0 --no-cpu-baseline --no-ess > gpurun_out/ws_${tag}.log 2>&1
python3 - <<PY
import csv, glob
tot = 0.0
for f in glob.glob("gpurun_out/ws_${tag}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rwm_block_kernel<7, false" in r["Kernel_Name"] and r["Counter_Name"] == "WRITE_SIZE":
            tot += float(r["Counter_Value"])
print("${tag}: WRITE_SIZE %.1f MB for ${chains} chains x 80 kB of trace: x %.2f" % (tot * 1024 / 1e6, tot * 1024 / (${chains} * 80000.0)))
PY
