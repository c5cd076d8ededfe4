#!/bin/bash
# config 3: serial (--opt streams=0) vs forked (1) 2D pass
for v in 0 1 0 1; do
  python bench.py --opt streams=$v --no-cpu-baseline --no-online --no-config5 2>/dev/null > gpurun_out/streams_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/streams_$v.json').read().strip().splitlines()[-1]); print('streams=$v', round(d['value']), d['ms_per_step'])"
done
