#!/bin/bash
# Round-3 per-kernel evidence on the GPU box (beside tools/r03_profile.sh): GEMM shapes against the platform library, fp8 shapes,
# cold weight-streaming shapes, the access-shape probe, attention shapes.   usage: bash tools/r03_evidence.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03
mkdir -p $OUT
timeout -k 10 300 python3 tools/gemm_ab.py -1 0 > $OUT/gemm_vs_library.txt 2>&1 || exit 1
echo "gemm_ab done"
timeout -k 10 300 python3 tools/gemm_shapes.py > $OUT/gemm_shapes_in_run.txt 2>&1 || exit 1
echo "gemm_shapes done"
timeout -k 10 200 python3 tools/fp8_bench.py > $OUT/fp8_shapes.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/stream_bench.py 0 blas > $OUT/stream_cold.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/attn_bench.py > $OUT/attn_shapes.txt 2>&1 || exit 1
echo "small benches done"
ls $OUT
