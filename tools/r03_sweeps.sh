#!/bin/bash
# Round-3 sweep evidence -> gpurun_out/$1/: per-config pass times with per-kernel microseconds (config 2, the per-rank tiles of a
# 2 / 4 / 8-GPU run of config 3, the k_c sweep, the 8-GPU tile of config 5), per-rank tile timings of the sharded choreography.
# Every run keeps its own stderr file.
TAG=${1:-r03sw}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
echo "bench.py --no-cpu-baseline --no-online --config <c> on one MI355X (per-kernel us from HIP events on the launch streams)" > $O/config_sweep.txt
for c in cfg2 cfg3_tile8 cfg3_tile4 cfg3_tile2 cfg3_kc8 cfg3_kc16 cfg5_tile8; do
  python3 bench.py --no-cpu-baseline --no-online --config $c 2> $O/sweep_$c.err | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
ks = ' '.join('{} {:.0f}'.format(k['name'], k['us']) for k in d['roofline'].get('kernels', []))
print('{:<12s} {:8.4f} ms/pass {:10.0f} subdomains/s   {}'.format('$c', d['ms_per_step'], d['value'], ks))
" >> $O/config_sweep.txt
done
for c in cfg3 cfg3_tile2 cfg3_tile4 cfg3_tile8; do echo "== $c"; python3 tools/phase_time.py $c 2>&1 | grep -v amdgpu.ids; done > $O/tile_times.txt
cat $O/config_sweep.txt; cat $O/tile_times.txt
