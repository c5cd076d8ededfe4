#!/bin/bash
# Config 5 (3D, P2) evidence into gpurun_out/cfg5/: kernel stats (rocprofv3 --kernel-trace --stats), bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5; rm -rf $O; mkdir -p $O
# --opt3 serial=1: the three chains of the pass on one stream, as in the bench's per-kernel table (overlapping kernels stretch
# each other's durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --config cfg5 --opt3 serial=1 --steps 5 --warmup 1 --no-cpu-baseline --no-online > $O/stats.log 2>&1
cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv
echo stats done
python3 bench.py --config cfg5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
