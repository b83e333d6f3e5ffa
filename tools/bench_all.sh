#!/bin/bash
# The round's record runs: every bench.py workload (with the CPU baseline leg) -> gpurun_out/<tag>_<wl>_bench.json,
# then the block / row kernel microbenchmarks.  Usage (GPU box): bash tools/bench_all.sh r3 [part]
#   part 1: c2 c3 c5 r32 r64      part 2: c4, c4s --fast, c4s (exact)      part 3: microbenchmarks     (default: all)
set -e
tag=$1
part=${2:-all}
cd "$(dirname "$0")/.."
out=gpurun_out
if [ "$part" = all ] || [ "$part" = 1 ]; then
    timeout -k 10 200 python3 bench.py > $out/${tag}_c2_bench.json
    timeout -k 10 200 python3 bench.py --workload c3 > $out/${tag}_c3_bench.json
    timeout -k 10 200 python3 bench.py --workload c5 > $out/${tag}_c5_bench.json
    timeout -k 10 200 python3 bench.py --workload r32 > $out/${tag}_r32_bench.json
    timeout -k 10 200 python3 bench.py --workload r64 > $out/${tag}_r64_bench.json
fi
if [ "$part" = all ] || [ "$part" = 2 ]; then
    timeout -k 10 300 python3 bench.py --workload c4 --steps 2 --warmup 1 > $out/${tag}_c4_bench.json
    timeout -k 10 300 python3 bench.py --workload c4 --fast --steps 2 --warmup 1 > $out/${tag}_c4_fast_bench.json
    timeout -k 10 300 python3 bench.py --workload c4s --fast --steps 2 --warmup 1 > $out/${tag}_c4s_fast_bench.json
    timeout -k 10 400 python3 bench.py --workload c4s --steps 2 --warmup 1 > $out/${tag}_c4s_bench.json
fi
if [ "$part" = all ] || [ "$part" = 3 ]; then
    : > $out/${tag}_microbench_block.txt
    for m in block block_mid shape mid; do
        echo "## tools/microbench.py $m" >> $out/${tag}_microbench_block.txt
        timeout -k 10 280 python3 tools/microbench.py $m 2>&1 | grep -v amdgpu.ids >> $out/${tag}_microbench_block.txt
    done
fi
echo done
