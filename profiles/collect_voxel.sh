#!/bin/bash
# Voxel stage (BASELINE config 2 shape: 10 M float64 points, voxel 0.2 m, 500 000-row chunks):
# kernel times, PMC traffic and one SQ pass in four separate rocprofv3 runs.   bash profiles/collect_voxel.sh <tag>
set -e -o pipefail
tag=$1
points=${2:-10000000}; voxel=${3:-0.2}; chunk=${4:-500000}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/stats --output-format csv -- python3 tools/voxel_probe.py $points $voxel $chunk > $out/stats.txt 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch --output-format csv -- python3 tools/voxel_probe.py $points $voxel $chunk > $out/fetch.txt 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write --output-format csv -- python3 tools/voxel_probe.py $points $voxel $chunk > $out/write.txt 2> $out/write.err
# one SQ pass (vector instructions per launch -> per row; wave cycles)
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS -d $out/pmc_sq1 --output-format csv -- python3 tools/voxel_probe.py $points $voxel $chunk > $out/sq1.txt 2> $out/sq1.err
python3 profiles/summarize.py $tag $out/stats $out/pmc_fetch $out/pmc_write $points voxel v${voxel}_chunk$chunk $out/pmc_sq1 > $out/summary.txt
cp $out/stats.txt profiles/${tag}_probe.txt
rm -rf $out/stats $out/pmc_fetch $out/pmc_write $out/pmc_sq1
echo "[collect_voxel $tag] done"
