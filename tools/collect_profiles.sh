#!/bin/bash
# Runs on the GPU box (gpurun): the rocprofv3 passes and bench lines that profiles/ keeps for a round.
#   bash tools/collect_profiles.sh <tag>        e.g. r02b
# kernel stats: one batch in flight, so that a kernel's duration is its own; PMC passes separate from --stats.
set -o pipefail
tag=${1:-r02}
out=gpurun_out/${tag}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
common="--steps 50 --warmup 10 --no-cpu-baseline --no-extras --inflight 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_small -o s -- python bench.py $common > $out/stats_small.json 2> $out/stats_small.err || exit 1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_small -o f -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 > /dev/null 2> $out/fetch_small.err || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_small -o w -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 > /dev/null 2> $out/write_small.err || exit 1
python tools/make_traffic_profile.py $out/fetch_small $out/write_small 256 $out/traffic_b256.json > /dev/null || exit 1
# PMC traffic of the other two measured configurations (missing item 4)
for v in full:512 valexnet:256; do
  name=${v%%:*}; b=${v##*:}
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch_$name -o f -- python bench.py --variant $name --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 > /dev/null 2> $out/fetch_$name.err || exit 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write_$name -o w -- python bench.py --variant $name --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-extras --inflight 1 > /dev/null 2> $out/write_$name.err || exit 1
  python tools/pmc_summary.py $out/fetch_$name > $out/traffic_${name}_fetch.txt || exit 1
  python tools/pmc_summary.py $out/write_$name > $out/traffic_${name}_write.txt || exit 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_full -o s -- python bench.py --variant full --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-extras --inflight 1 > $out/stats_full.json 2> $out/stats_full.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_valexnet -o s -- python bench.py --variant valexnet --batch 256 --steps 50 --warmup 10 --no-cpu-baseline --no-extras --inflight 1 > $out/stats_valexnet.json 2> $out/stats_valexnet.err || exit 1
python bench.py > $out/bench_small.json 2> $out/bench_small.err || exit 1
python bench.py --variant full --steps 10 --warmup 3 > $out/bench_full.json 2> $out/bench_full.err || exit 1
python bench.py --variant valexnet > $out/bench_valexnet.json 2> $out/bench_valexnet.err || exit 1
python bench.py --input u8 --no-cpu-baseline --no-extras > $out/bench_small_u8.json 2> $out/bench_small_u8.err || exit 1
ls $out
