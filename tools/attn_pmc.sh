#!/bin/bash
# Counter passes for the ViT attention kernel (run on the GPU box from the repo root): bash tools/attn_pmc.sh <tag> [mode]
set -e
TAG=${1:-r03}; MODE=${2:-0}
OUT=gpurun_out/$TAG/attn_pmc
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/attn_pmc.py $MODE 10 > $OUT/timing_mode$MODE.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/attn_pmc.py $MODE 5 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o p -- python3 tools/attn_pmc.py $MODE 3 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p2 -o p -- python3 tools/attn_pmc.py $MODE 3 > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p -- python3 tools/attn_pmc.py $MODE 3 > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p4 -o p -- python3 tools/attn_pmc.py $MODE 3 > $OUT/p4.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p5 -o p -- python3 tools/attn_pmc.py $MODE 3 > $OUT/p5.log 2>&1
python3 tools/attn_pmc_summary.py $OUT > $OUT/summary_mode$MODE.txt
cat $OUT/summary_mode$MODE.txt
