#!/bin/bash
# Kernel-trace stats of one bench workload on the GPU box: bash tools/kt_workload.sh <workload> <tag> [extra bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W=$1; TAG=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline "$@" > $OUT/kt_$W.log 2>&1 || exit 1
python3 tools/summarize_rocprof.py $OUT/kt_$W $OUT/${W}_kernel_stats.md "bench.py --workload $W --steps 5 --warmup 2 $* on 1xMI355X" > /dev/null
cp $OUT/kt_$W/*/*kernel_stats.csv $OUT/${W}_kernel_stats.csv 2>/dev/null
rm -rf $OUT/kt_$W
tail -2 $OUT/kt_$W.log | cut -c1-400
