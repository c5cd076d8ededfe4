#!/bin/bash
# config 3: serial (LRBMS_STREAMS=0) vs forked (1) 2D pass
for v in 0 1 0 1; do
  LRBMS_STREAMS=$v python bench.py --no-cpu-baseline --no-online --no-config5 2>/dev/null > gpurun_out/streams_$v.json
  python -c "
import json; d=json.loads(open('gpurun_out/streams_$v.json').read().strip().splitlines()[-1]); print('LRBMS_STREAMS=$v', round(d['value']), d['ms_per_step'])"
done
