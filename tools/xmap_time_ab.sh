export TMPDIR=/tmp
for m in 0 auto 0 auto; do
  if [ $m = auto ]; then unset ORE_XMAP; else export ORE_XMAP=$m; fi
  d=gpurun_out/ab_$m$RANDOM
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 tools/protocol_loop.py 300 > /dev/null 2>&1 || exit 1
  python3 tools/trace_summary.py $d 30 > $d.txt
  python3 tools/conv_layers_table.py $d.txt $d.layers.txt
  echo "xmap=$m $(awk '{printf "%s ", $(NF-3)}' $d.layers.txt | cut -d' ' -f4-)"
  rm -rf $d
done
