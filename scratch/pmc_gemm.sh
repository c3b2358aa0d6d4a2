#!/bin/bash
# usage: pmc_gemm.sh M N K which tag
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
M=$1; N=$2; K=$3; W=$4; TAG=$5
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  NAME=$(echo $C | tr ' ' '_')
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_${TAG}_${NAME} -- python3 scratch/gemm_one.py $M $N $K $W > gpurun_out/pmc_${TAG}_${NAME}.log 2>&1
  echo "pass $NAME rc=$?"
  grep -h gemm_bf16 gpurun_out/pmc_${TAG}_${NAME}/*/*counter_collection.csv 2>/dev/null | awk -F, '{print $(NF-1), $NF}' | sort | uniq -c | head -8
done
