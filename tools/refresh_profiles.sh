#!/bin/bash
# Regenerate the committed measurement artifacts of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/refresh_profiles.sh r02
# Writes everything under gpurun_out/<tag>/; copy what should be judged into profiles/ (the last lines print the cp commands).
set -e -o pipefail
TAG=${1:-r02}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 1. PMC traffic first (bench.py reads profiles/<tag>_pmc_traffic.json for roofline.traffic): two separate counter passes
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_pass.py > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_pass.py > $OUT/pmc_write.log 2>&1
VER=$(python3 -c "import sys; sys.path.insert(0, 'faster-orefsdet_amd'); import orehip; print(orehip.lib().ore_version())")
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json $VER > $OUT/${TAG}_pmc_traffic.txt
cp $OUT/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json
ORE_OPERANDS=bf16s timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_s -- python3 tools/pmc_pass.py > $OUT/pmc_fetch_s.log 2>&1
ORE_OPERANDS=bf16s timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_s -- python3 tools/pmc_pass.py > $OUT/pmc_write_s.log 2>&1
python3 tools/pmc_summary.py $OUT/pmc_fetch_s $OUT/pmc_write_s $OUT/${TAG}_pmc_traffic_bf16s.json $VER > $OUT/${TAG}_pmc_traffic_bf16s.txt
cp $OUT/${TAG}_pmc_traffic_bf16s.json profiles/${TAG}_pmc_traffic_bf16s.json
# 2. one image of the headline protocol, kernel by kernel, and the per-layer conv table
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proto -o proto -- python3 tools/protocol_loop.py 300 > $OUT/proto.log 2>&1
cp $(ls $OUT/proto/proto_kernel_stats.csv $OUT/proto/*/proto_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_protocol_kernel_stats.csv
python3 tools/trace_summary.py $OUT/proto 30 > $OUT/${TAG}_bench_image_timeline.txt
python3 tools/conv_layers_table.py $OUT/${TAG}_bench_image_timeline.txt $OUT/${TAG}_conv_layers.txt $VER
cp $OUT/${TAG}_conv_layers.json profiles/${TAG}_conv_layers.json
# 3b. the bs-16 training step kernel by kernel, fp32 and bf16 (bench.py reads the summaries into the train legs' roofline)
for PR in fp32 bf16; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_$PR -o train -- python3 tools/bench_train.py --batch 16 --steps 5 --warmup 2 --precision $PR > $OUT/train_$PR.log 2>&1
  python3 tools/stats_top.py $(ls $OUT/train_$PR/train_kernel_stats.csv $OUT/train_$PR/*/train_kernel_stats.csv 2>/dev/null | head -1) 7 45 $OUT/${TAG}_train_step_bs16_${PR}_summary.json $VER > $OUT/${TAG}_train_step_bs16_${PR}_kernel_stats.txt
  cp $OUT/${TAG}_train_step_bs16_${PR}_summary.json profiles/${TAG}_train_step_bs16_${PR}_summary.json
  rm -rf $OUT/train_$PR
done
# 3b'. the bs-1 step (graph-captured dense part), the per-call conv tables, the ROIAlign-backward forms, the box's MFMA / HBM ceilings
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bs1 -o train -- python3 tools/bench_train.py --batch 1 --steps 20 --warmup 5 > $OUT/train_bs1.log 2>&1
python3 tools/stats_top.py $(ls $OUT/train_bs1/train_kernel_stats.csv $OUT/train_bs1/*/train_kernel_stats.csv 2>/dev/null | head -1) 25 45 > $OUT/${TAG}_train_step_bs1_kernel_stats.txt
rm -rf $OUT/train_bs1
timeout -k 10 200 python3 tools/train_conv_table.py > $OUT/${TAG}_train_conv_table_bs16_fp32.txt 2> /dev/null
timeout -k 10 200 python3 tools/train_conv_table.py --precision bf16 > $OUT/${TAG}_train_conv_table_bs16_bf16.txt 2> /dev/null
timeout -k 10 200 python3 tools/train_glue_probe.py > $OUT/${TAG}_train_glue_aten.txt 2> /dev/null
timeout -k 10 200 python3 tools/roi_bwd_bench.py > $OUT/${TAG}_roi_bwd_tile.txt 2> /dev/null
timeout -k 10 100 python3 tools/hbm_bw_probe.py 2> /dev/null | grep -v amdgpu.ids > $OUT/${TAG}_hbm_bw.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_peak tools/mfma_peak.hip && timeout -k 10 100 /tmp/mfma_peak > $OUT/${TAG}_mfma_peak.txt
# 3c. the default bench command under the kernel trace (the judged line + its rocprof summary)
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cp $(ls $OUT/bench/bench_kernel_stats.csv $OUT/bench/*/bench_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_bench_kernel_stats.csv
# 4. the same without the profiler (the numbers quoted in DESIGN.md), fp32 and bf16-operand mode
timeout -k 10 300 python3 bench.py > $OUT/${TAG}_bench_noprof.json 2> $OUT/bench_noprof.err
timeout -k 10 300 python3 bench.py --conv-operands bf16s --no-cpu-baseline --no-train-leg > $OUT/${TAG}_bench_bf16s.json 2> $OUT/bench_bf16s.err
# 5. the bf16-storage mode kernel by kernel
ORE_OPERANDS=bf16s timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/proto_s -o proto -- python3 tools/protocol_loop.py 300 > $OUT/proto_s.log 2>&1
python3 tools/trace_summary.py $OUT/proto_s 30 > $OUT/${TAG}_bench_image_timeline_bf16s.txt
python3 tools/conv_layers_table.py $OUT/${TAG}_bench_image_timeline_bf16s.txt $OUT/${TAG}_conv_layers_bf16s.txt $VER
cp $OUT/${TAG}_conv_layers_bf16s.json profiles/${TAG}_conv_layers_bf16s.json
rm -rf $OUT/proto $OUT/proto_s $OUT/bench $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_s $OUT/pmc_write_s
tail -3 $OUT/${TAG}_conv_layers.txt
echo "done: $OUT"
