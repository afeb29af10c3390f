#!/bin/bash
# Profiles of one round (run on the GPU box through gpurun): kernel-trace stats of bench.py (default: three contexts in flight; and
# --in-flight 1: every kernel alone on the chip, the per-kernel averages), HBM traffic counters and SQ
# counters of devtools/pmc_target.py. Usage: bash devtools/gpu_profile_job.sh <tag>   -> gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --cpu-seconds 0 --no-single-image --no-cli --no-pmc > $OUT/bench_under_rocprof.json 2> $OUT/stats.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats1 -o bench1 -- python3 $R/bench.py --in-flight 1 --cpu-seconds 0 --no-single-image --no-standalone --no-cli --no-pmc > $OUT/bench1_under_rocprof.json 2> $OUT/stats1.log &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $R/devtools/pmc_target.py > $OUT/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $R/devtools/pmc_target.py > $OUT/pmc_write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq1 -o pmc -- python3 $R/devtools/pmc_target.py > $OUT/pmc_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $R/devtools/pmc_target.py > $OUT/pmc_sq2.log 2>&1
rc=$?
find $OUT -name "*.csv" | head -20
exit $rc
