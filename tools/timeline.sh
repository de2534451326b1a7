#!/bin/bash
# One image of the headline protocol kernel by kernel + the per-layer conv table (step 3 of refresh_profiles.sh alone):
#   bash tools/timeline.sh r03a      -> gpurun_out/<tag>/<tag>_bench_image_timeline.txt, <tag>_conv_layers.txt
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proto -o proto -- python3 tools/protocol_loop.py 300 > $OUT/proto.log 2>&1
cp $(ls $OUT/proto/proto_kernel_stats.csv $OUT/proto/*/proto_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_protocol_kernel_stats.csv
python3 tools/trace_summary.py $OUT/proto 30 > $OUT/${TAG}_bench_image_timeline.txt
python3 tools/conv_layers_table.py $OUT/${TAG}_bench_image_timeline.txt $OUT/${TAG}_conv_layers.txt
rm -rf $OUT/proto
cat $OUT/${TAG}_conv_layers.txt
