#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes (separate, as the microarchitecture guide prescribes) over one bench workload, summed per launch for
# the kernels whose name contains <substring>: bash tools/pmc_workload.sh <workload> <kernel substring> <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W=$1; KERNEL=$2; TAG=$3
OUT=gpurun_out/$TAG
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc_${W}_$C -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline > $OUT/pmc_${W}_$C.log 2>&1 || exit 1
done
python3 tools/pmc_summary.py $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE $KERNEL $OUT/pmc_${W}_${KERNEL}.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --workload $W --steps 2 --warmup 1; all $KERNEL launches of the run"
rm -rf $OUT/pmc_${W}_FETCH_SIZE $OUT/pmc_${W}_WRITE_SIZE
