#!/bin/bash
# Round-4 kernel-trace summaries of single workloads on the GPU box (the program goes directly after `--`).
# usage (from the repo root, on the box):  bash tools/r04_kt.sh <commit> <workload> [<workload> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
COMMIT=${1:-unknown}; shift
OUT=gpurun_out/r04
mkdir -p $OUT
for W in "$@"; do
  rm -rf $OUT/kt_$W
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$W -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline > $OUT/kt_$W.log 2>&1 || { echo "$W: rocprofv3 failed"; tail -5 $OUT/kt_$W.log; continue; }
  python3 tools/summarize_rocprof.py $OUT/kt_$W $OUT/${W}_kernel_stats.md "bench.py --workload $W --steps 5 --warmup 2 (7 steps) on 1xMI355X, build at $COMMIT" > /dev/null
  rm -rf $OUT/kt_$W
  echo "$W kernel trace done"
done
ls $OUT
