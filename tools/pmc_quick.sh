set -e -o pipefail
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/pmc_pass.py > $OUT/pmc_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/pmc_pass.py > $OUT/pmc_write.log 2>&1
VER=$(python3 -c "import sys; sys.path.insert(0, 'faster-orefsdet_amd'); import orehip; print(orehip.lib().ore_version())")
python3 tools/pmc_summary.py $OUT/pmc_fetch $OUT/pmc_write $OUT/${TAG}_pmc_traffic.json $VER > $OUT/${TAG}_pmc_traffic.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write
cat $OUT/${TAG}_pmc_traffic.txt
