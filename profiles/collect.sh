#!/bin/bash
# Collects one profile set on the GPU box (five separate rocprofv3 runs, as MI355X_MICROARCH.md
# prescribes: kernel times and PMC counters never in the same run; program directly after "--").
#   bash profiles/collect.sh <tag> <points> <kind> <frame>
# Output: gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_sq1,pmc_sq2} + profiles/<tag>_*.csv + traffic.json entry.
set -e -o pipefail
tag=$1; points=$2; kind=$3; frame=$4
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="bench.py --points $points --kind $kind --frame $frame --no-cpu-baseline --no-tile-stream --no-side"
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 $args --steps 5 --warmup 1 > $out/stats.json 2> $out/stats.err
echo "[collect $tag] stats done"
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 $args --steps 2 --warmup 1 > $out/fetch.json 2> $out/fetch.err
echo "[collect $tag] fetch done"
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 $args --steps 2 --warmup 1 > $out/write.json 2> $out/write.err
echo "[collect $tag] write done"
# SQ counters (VALU issue, wave cycles): two more counters-only passes
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d $out/pmc_sq1 --output-format csv -- python3 $args --steps 2 --warmup 1 > $out/sq1.json 2> $out/sq1.err
echo "[collect $tag] sq1 done"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $out/pmc_sq2 --output-format csv -- python3 $args --steps 2 --warmup 1 > $out/sq2.json 2> $out/sq2.err
echo "[collect $tag] sq2 done"
python3 profiles/summarize.py $tag $out/stats $out/pmc_fetch $out/pmc_write $points $kind $frame $out/pmc_sq1 $out/pmc_sq2 > $out/summary.txt
cp $out/stats.json profiles/${tag}_bench.json
# the raw rocprof directories are large: keep only the summaries in gpurun_out
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_sq1 $out/pmc_sq2
echo "[collect $tag] summarised"
