#!/bin/bash
# Collects the round's final evidence into gpurun_out/final/: kernel stats, PMC traffic (separate passes), bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-online > $O/stats.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-online > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-online > $O/write.log 2>&1
echo write done
python3 tools/pmc_traffic.py 3 $O/fetch $O/write > $O/pmc_traffic.txt 2>&1
cat $O/pmc_traffic.txt
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json
