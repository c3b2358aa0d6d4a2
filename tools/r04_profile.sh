#!/bin/bash
# Round-4 evidence run on the GPU box: headline bench line, kernel-trace stats of the product configuration (two batch slices on two
# streams) and of the serial configuration the roofline is measured in, PMC traffic passes, bench lines of the other workloads.
# usage (from the repo root, on the box):  bash tools/r04_profile.sh <commit> [quick]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
COMMIT=${1:-unknown}
QUICK=${2:-}
OUT=gpurun_out/r04
mkdir -p $OUT
if [ "$QUICK" = "partB" ]; then
for W in idefics9b_train_bs8 idefics9b_train_bs8_cached_vision idefics9b_train_bs8_cached_teacher idefics9b_generate_bs8 idefics9b_student_bs8 idefics2_8b_1shot_bs8 idefics2_8b_32shot_bs8 idefics2_8b_32shot_fp8_bs8 frontend_images_bs8; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 8 --warmup 3 > $OUT/bench_$W.json.log 2>$OUT/bench_$W.err
  echo "$W rc=$?"
done
ls $OUT; exit 0; fi
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline --batch-streams 1 > $OUT/kt.log 2>&1 || exit 1
python3 tools/summarize_rocprof.py $OUT/kt $OUT/bench_headline_serial_kernel_stats.md "bench.py --steps 5 --warmup 2 --batch-streams 1 (7 forwards, one stream: the configuration roofline.achieved is measured in) on 1xMI355X, headline workload, build at $COMMIT" > /dev/null
echo "serial kernel trace done"
if [ "$QUICK" = "quick" ]; then cp $OUT/kt/*/*kernel_stats.csv $OUT/kernel_stats_full.csv 2>/dev/null; rm -rf $OUT/kt; ls $OUT; exit 0; fi
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_headline.json.log 2>$OUT/bench_headline.err || exit 1
echo "headline done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt2 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gpu-baseline --no-profiler > $OUT/kt2.log 2>&1 || exit 1
python3 tools/summarize_rocprof.py $OUT/kt2 $OUT/bench_headline_sliced_kernel_stats.md "bench.py --steps 5 --warmup 2 --no-profiler (7 forwards, product configuration: two batch slices on two HIP streams, native layer runner; a kernel's duration includes the time the other slice's kernels held CUs) on 1xMI355X, build at $COMMIT" > /dev/null
echo "kernel traces done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline --batch-streams 1 > $OUT/pmc_$C.log 2>&1 || exit 1
  echo "pmc $C done"
done
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE gemm_bf16_ $OUT/pmc_headline_gemm.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --batch-streams 1; all gemm_bf16_* launches of the run (flow64, quad64, 128-tile)" > /dev/null
python3 - <<PY
import json
p="$OUT/pmc_headline_gemm.json"; d=json.load(open(p)); d["commit"]="$COMMIT"; json.dump(d, open(p,"w"), indent=1)
PY
if [ "$QUICK" = "partA" ]; then rm -rf $OUT/kt $OUT/kt2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE; ls $OUT; exit 0; fi
for W in idefics9b_train_bs8 idefics9b_train_bs8_cached_vision idefics9b_train_bs8_cached_teacher idefics9b_generate_bs8 idefics9b_student_bs8 idefics2_8b_1shot_bs8 idefics2_8b_32shot_bs8 idefics2_8b_32shot_fp8_bs8 frontend_images_bs8; do
  timeout -k 10 400 python3 bench.py --workload $W --steps 8 --warmup 3 > $OUT/bench_$W.json.log 2>$OUT/bench_$W.err
  echo "$W rc=$?"
done
rm -rf $OUT/kt $OUT/kt2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls $OUT
