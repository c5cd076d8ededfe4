#!/bin/bash
# Round-2 evidence set -> gpurun_out/$1/: kernel stats, PMC traffic (FETCH / WRITE in separate passes), SQ counters
# (MFMA busy, issue stalls), bench line.  Program directly after `--` (no env / bash -c hop under the profiler).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-online"
rocprofv3 -L > $O/counters_available.txt 2>&1
grep -i -E "MFMA|SQ_BUSY|SQ_WAVE_CYCLES|SQ_WAIT|SQ_ACTIVE_INST|SQ_INSTS_VALU " $O/counters_available.txt | cut -c1-160 > $O/counters_mfma.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/stats.log 2>&1 && cp $O/stats/*/*kernel_stats.csv $O/kernel_stats.csv && echo stats done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 && echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 && echo write done
python3 tools/pmc_traffic.py cfg3 $O/pmc_traffic.json $O/fetch $O/write > $O/pmc_traffic.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1 && echo sq1 done
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 && echo sq2 done
python3 tools/pmc_sq.py $O/sq1 $O/sq2 > $O/pmc_sq.txt 2>&1
tail -n 20 $O/pmc_traffic.txt
cut -c1-400 $O/pmc_sq.txt | head -40
# keep only the summaries (the raw traces are large)
rm -rf $O/stats $O/fetch $O/write $O/sq1 $O/sq2
