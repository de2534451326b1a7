set -e -o pipefail
TAG=r05; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py > $OUT/${TAG}_bench.json 2> $OUT/bench.err
cp $(ls $OUT/bench/bench_kernel_stats.csv $OUT/bench/*/bench_kernel_stats.csv 2>/dev/null | head -1) $OUT/${TAG}_bench_kernel_stats.csv
timeout -k 10 300 python3 bench.py > $OUT/${TAG}_bench_noprof.json 2> $OUT/bench_noprof.err
timeout -k 10 300 python3 bench.py --conv-operands bf16s --no-cpu-baseline --no-train-leg > $OUT/${TAG}_bench_bf16s.json 2> $OUT/bench_bf16s.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train_bs1g -o train -- python3 tools/bench_train.py --batch 1 --steps 40 --warmup 5 --graph-step > $OUT/train_bs1g.log 2>&1
python3 tools/stats_top.py $(ls $OUT/train_bs1g/train_kernel_stats.csv $OUT/train_bs1g/*/train_kernel_stats.csv 2>/dev/null | head -1) 45 45 > $OUT/${TAG}_train_step_bs1_graph_kernel_stats.txt
rm -rf $OUT/bench $OUT/train_bs1g
echo done
