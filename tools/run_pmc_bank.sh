#!/bin/bash
# Both PMC passes (separate, counters only with --kernel-trace) for the bank pass at the fp32 bench size and at the
# config-5 size (bf16 banks, K=131072); writes gpurun_out/pmc/*.json.  Run on the GPU box from the repo root.
set -e
R=$PWD
mkdir -p $R/gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
for cfg in "16384 131072 f32" "131072 131072 bf16"; do
  set -- $cfg
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$ctr
    K=$1 N=$2 DTYPE=$3 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -o p -- python $R/tools/pmc_bank.py > /dev/null 2>&1
  done
  python $R/tools/pmc_bank_json.py $R/gpurun_out/pmc/bank_pass_pmc_$3_K$1.json $(find /tmp/pmc_FETCH_SIZE -name '*counter_collection.csv') $(find /tmp/pmc_WRITE_SIZE -name '*counter_collection.csv') $1 $2 $3
done
