#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5fetch; rm -rf $O; mkdir -p $O
B="python3 bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-online"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 && echo fetch done
python3 tools/pmc3d.py $O/fetch | awk '{print $1, $NF, $(NF-1), $(NF-2)}'
rm -rf $O/fetch
