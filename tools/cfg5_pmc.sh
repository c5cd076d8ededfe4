#!/bin/bash
# PMC passes of the config-5 bench (separate passes, kernel trace only) into gpurun_out/cfg5pmc/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cfg5pmc; rm -rf $O; mkdir -p $O
B="python3 bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-online"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1 && echo fetch done
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1 && echo write done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1 && echo sq1 done
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1 && echo sq2 done
python3 tools/pmc3d.py $O/fetch $O/write > $O/traffic.txt 2>&1
python3 tools/pmc3d.py $O/sq1 $O/sq2 > $O/sq.txt 2>&1
cut -c1-60,60- $O/traffic.txt | awk '{print $1, $NF, $(NF-1), $(NF-2), $(NF-3)}'
rm -rf $O/fetch $O/write $O/sq1 $O/sq2
