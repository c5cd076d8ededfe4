#!/bin/bash
# bench.py's N > 1 path at config 3's own sizes with N ranks sharing cuda:0 over gloo (halo rows staged through the host): the
# checksums of the N-rank run against the one-rank run.  usage (GPU box): tools/bench_ranks_rehearsal.sh [N=4]   (N <= 6)
N=${1:-4}
export LRBMS_BENCH_BACKEND=gloo LRBMS_BENCH_DEVICE=0
A="--steps 3 --warmup 1 --config cfg3 --no-cpu-baseline --no-online --no-config5"
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29917 bench.py --gpus $N $A 2>/dev/null | grep '^{' > /tmp/ranksN.json
python3 bench.py --gpus 1 $A 2>/dev/null | grep '^{' > /tmp/ranks1.json
python3 - <<PY
import json
a = json.load(open('/tmp/ranksN.json')); b = json.load(open('/tmp/ranks1.json'))
print('ranks', a['n_gpus'], 'ms_per_step', round(a['ms_per_step'], 4), '| one rank', round(b['ms_per_step'], 4))
print('checksum', a['output_checksum'], b['output_checksum'])
print('abs     ', a['output_abs_checksum'], b['output_abs_checksum'])
rel = max(abs(a['output_checksum'] - b['output_checksum']), abs(a['output_abs_checksum'] - b['output_abs_checksum'])) / b['output_abs_checksum']
print('relative difference', rel)
assert rel <= 1e-10, rel
print(a['distributed'])
PY
