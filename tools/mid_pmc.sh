#!/bin/bash
# Counter passes for the M = 256 GEMM kernel on one cold shape (run on the GPU box from the repo root):
#   bash tools/mid_pmc.sh <tag> <kernel substring> M N K
set -e
TAG=$1; KERNEL=$2; M=$3; N=$4; K=$5
OUT=gpurun_out/$TAG/mid_pmc_${M}_${N}_${K}
mkdir -p $OUT
export TMPDIR=/tmp
python3 tools/mid_pmc.py $M $N $K 3 > $OUT/timing.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 tools/mid_pmc.py $M $N $K 2 > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -o p -- python3 tools/mid_pmc.py $M $N $K 1 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/p2 -o p -- python3 tools/mid_pmc.py $M $N $K 1 > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -o p -- python3 tools/mid_pmc.py $M $N $K 1 > $OUT/p3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p5 -o p -- python3 tools/mid_pmc.py $M $N $K 1 > $OUT/p5.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/p6 -o p -- python3 tools/mid_pmc.py $M $N $K 1 > $OUT/p6.log 2>&1 || echo "p6 counters unavailable"
python3 tools/attn_pmc_summary.py $OUT $KERNEL > $OUT/summary.txt
cat $OUT/timing.txt $OUT/summary.txt
