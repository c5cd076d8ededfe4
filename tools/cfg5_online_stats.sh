#!/bin/bash
# rocprofv3 kernel statistics of the config-5 run INCLUDING the online phase and the snapshot solve -> gpurun_out/cfg5_online/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5_online; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline > $O/stats.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
head -30 $O/kernel_stats.csv
