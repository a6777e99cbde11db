#!/bin/bash
# Runs on the GPU box (gpurun): SQ counters per kernel of the TT-small forward at B = 256 (one batch in flight).
#   bash tools/collect_counters.sh <tag> [variant]  ->  gpurun_out/<tag>/counters_<variant>_b<batch>.json  (small: B = 256, full: 512)
# PMC passes are separate from --stats and carry no other trace domain (gpurun refuses the combination).
set -o pipefail
tag=${1:-r03}
variant=${2:-small}
batch=256; [ "$variant" = full ] && batch=512
export COUNTERS_BATCH=$batch
out=gpurun_out/${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--steps 5 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 --variant $variant ${BENCH_EXTRA}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES \
  --output-format csv -d $out/pmc_sq1_$variant -o c -- python bench.py $args > /dev/null 2> $out/pmc_sq1_$variant.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $out/pmc_sq2_$variant -o c -- python bench.py $args > /dev/null 2> $out/pmc_sq2_$variant.err || exit 1
python tools/make_counters_profile.py $out/counters_${variant}_b${batch}.json $out/pmc_sq1_$variant $out/pmc_sq2_$variant || exit 1
