#!/bin/bash
# Round-2 sweep evidence -> gpurun_out/$1/: full bench line (CPU baseline + online), per-config pass times with per-kernel
# microseconds, per-rank tile timings of the sharded choreography.
TAG=${1:-r02sw}
cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "bench.py --no-cpu-baseline --no-online --config <c> on one MI355X (per-kernel us from HIP events on the launch streams)" > $O/config_sweep.txt
for c in cfg2 cfg3_tile8 cfg3_tile4 cfg3_tile2 cfg3_kc8 cfg3_kc16; do
  python3 bench.py --no-cpu-baseline --no-online --config $c 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
ks = ' '.join('{} {:.0f}'.format(k['name'], k['us']) for k in d['roofline'].get('kernels', []))
print('{:<12s} {:8.4f} ms/pass {:10.0f} subdomains/s   {}'.format('$c', d['ms_per_step'], d['value'], ks))
" >> $O/config_sweep.txt
done
for c in cfg3 cfg3_tile2 cfg3_tile4 cfg3_tile8; do echo "== $c"; python3 tools/phase_time.py $c 2>&1 | grep -v amdgpu.ids; done > $O/tile_times.txt
cat $O/config_sweep.txt
tail -c 600 $O/bench.json
