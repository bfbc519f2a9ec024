#!/bin/bash
# Usage (on the GPU box, from the repo root): bash profiles/run_profile.sh <tag>
# Collects: kernel-trace stats, then three PMC passes (SQ busy/wait, FETCH_SIZE, WRITE_SIZE) of a 1-step bench.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
# (the self-certifying extras of bench.py — its own rocprofv3 child passes, the fp32 block, the oracle check — stay off under the profiler)
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-fp32-mode --no-parity --no-power --no-latency --no-configs --sustained-seconds 0 $BENCH_EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $BENCH > $OUT/trace.log 2>&1 || exit 1
BENCH1="python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pmc --no-fp32-mode --no-parity --no-power --no-latency --no-configs --sustained-seconds 0 $BENCH_EXTRA"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o p -- $BENCH1 > $OUT/pmc_sq.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o p -- $BENCH1 > $OUT/pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o p -- $BENCH1 > $OUT/pmc_write.log 2>&1 || exit 4
ls -R $OUT | head -30
