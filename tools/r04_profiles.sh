#!/bin/bash
# Round-4 evidence set -> gpurun_out/$1/ (copy the summaries to profiles/r04_*): for config 3 and config 5 each the kernel stats
# (rocprofv3 --kernel-trace --stats), PMC traffic (FETCH_SIZE / WRITE_SIZE in separate passes, JSON with the kernel-source hash for
# bench.py), SQ counters (MFMA busy, issue stalls), and the bench lines.  Program directly after `--` (no env / bash -c hop).
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B3="python3 bench.py --no-cpu-baseline --no-online --no-config5"      # the default 20 steps + 3 warm-up passes: steady-state averages
B5="python3 bench.py --config cfg5 --opt3 serial=1 --no-cpu-baseline --no-online"
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES"
for cfg in cfg3 cfg5; do
  if [ $cfg = cfg3 ]; then B=$B3; else B=$B5; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${cfg}_stats -- $B > $O/${cfg}_stats.log 2>&1 && cp $O/${cfg}_stats/*/*kernel_stats.csv $O/${cfg}_kernel_stats.csv && echo $cfg stats done
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${cfg}_fetch -- $B > $O/${cfg}_fetch.log 2>&1 && echo $cfg fetch done
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${cfg}_write -- $B > $O/${cfg}_write.log 2>&1 && echo $cfg write done
  rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/${cfg}_sq1 -- $B > $O/${cfg}_sq1.log 2>&1 && echo $cfg sq1 done
  rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/${cfg}_sq2 -- $B > $O/${cfg}_sq2.log 2>&1 && echo $cfg sq2 done
done
python3 tools/pmc_traffic.py cfg3 $O/cfg3_pmc_traffic.json $O/cfg3_fetch $O/cfg3_write > $O/cfg3_pmc_traffic.txt 2>&1
python3 tools/pmc_sq.py $O/cfg3_sq1 $O/cfg3_sq2 > $O/cfg3_pmc_sq.txt 2>&1
python3 tools/pmc3d.py --json $O/cfg5_pmc_traffic.json cfg5 $O/cfg5_fetch $O/cfg5_write > $O/cfg5_pmc_traffic.txt 2>&1
python3 tools/pmc3d.py $O/cfg5_sq1 $O/cfg5_sq2 > $O/cfg5_pmc_sq.txt 2>&1
rm -rf $O/*_stats $O/*_fetch $O/*_write $O/*_sq1 $O/*_sq2
for c in cfg3 cfg3_tile2 cfg3_tile4 cfg3_tile8; do echo "== $c"; python3 tools/phase_time.py $c 2> /dev/null; done > $O/tile_times.txt
for c in cfg2 cfg3 cfg3_tile8 cfg3_tile4 cfg3_tile2 cfg3_kc8 cfg3_kc16; do python3 tools/kernel_times.py $c 2> /dev/null; done > $O/config_sweep.txt
python3 tools/online_time.py 16 32 64 > $O/online_time.txt 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/online_stats -- python3 tools/online_time.py 64 > $O/online_stats.log 2>&1 && cp $O/online_stats/*/*kernel_stats.csv $O/online_kernel_stats.csv; rm -rf $O/online_stats
python3 bench.py > $O/bench.json 2> $O/bench.err
python3 bench.py --config cfg5 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err
tail -n 12 $O/cfg3_pmc_traffic.txt; tail -c 300 $O/bench.json
