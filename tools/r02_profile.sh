#!/bin/bash
# Round-2 evidence run on the GPU box: kernel-trace stats + PMC traffic passes of the headline bench, bench lines of every workload.
# usage (from the repo root, on the box):  bash tools/r02_profile.sh <commit>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
COMMIT=${1:-unknown}
OUT=gpurun_out/r02
mkdir -p $OUT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_headline.json.log 2>$OUT/bench_headline.err || exit 1
echo "headline done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline > $OUT/kt.log 2>&1 || exit 1
python3 tools/summarize_rocprof.py $OUT/kt $OUT/bench_headline_kernel_stats.md "bench.py --steps 5 --warmup 2 (7 forwards) on 1xMI355X, headline workload, round-2 build at $COMMIT" > /dev/null
echo "kernel trace done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline > $OUT/pmc_$C.log 2>&1 || exit 1
  echo "pmc $C done"
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE gemm_bf16_ $OUT/pmc_headline_gemm.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1; all gemm_bf16_* launches of the run (ping-pong, flow, 128-tile)" > /dev/null
python3 - <<PY
import json
p="$OUT/pmc_headline_gemm.json"; d=json.load(open(p)); d["commit"]="$COMMIT"; json.dump(d, open(p,"w"), indent=1)
PY
for W in idefics9b_train_bs8 idefics9b_generate_bs8 idefics9b_student_bs8 idefics2_8b_1shot_bs8 idefics2_8b_32shot_bs8 idefics2_8b_32shot_fp8_bs8; do
  timeout -k 10 300 python3 bench.py --workload $W --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_$W.json.log 2>$OUT/bench_$W.err
  echo "$W rc=$?"
done
rm -rf $OUT/kt/*/*kernel_trace.csv $OUT/pmc_*/*/*kernel_trace.csv $OUT/pmc_*/*/*counter_collection.csv
ls $OUT
