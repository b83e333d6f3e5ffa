#!/bin/bash
# tuning: C5 shape with 1..7 dot waves per chain group (MCMC_GIBBS_NDOT), 1/4 of the chains
set -e
cd "$(dirname "$0")/.."
for nd in 1 2 3 5 7; do
  MCMC_GIBBS_NDOT=$nd timeout -k 10 120 python bench.py --workload c5 --chains 262144 --steps 2 --warmup 1 --no-cpu-baseline --no-ess 2>&1 | tail -1 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('ndot=$nd', round(l['ms_per_step'],1), 'ms', '%.3e'%l['value'], l['rhat_max'], l['chains_failed'])"
done
