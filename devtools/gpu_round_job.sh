#!/bin/bash
# The round's bench lines (run on the GPU box through gpurun): C4 in full (roofline, cli, cpu baseline), the other BASELINE GPU
# configurations and the reference's own, a lone one-stream context under the kernel tracer (every kernel alone on the chip) and
# the timelines of single images. Usage: bash devtools/gpu_round_job.sh <tag>   -> gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R &&
python bench.py > $OUT/bench_C4.json 2> $OUT/bench_C4.err &&
for w in C2 C3 C5 REF; do python bench.py --workload $w --cpu-seconds 3 --no-standalone --no-cli > $OUT/bench_$w.json 2>> $OUT/bench_other.err || exit 1; done
cd /tmp && export TMPDIR=/tmp &&
MUSICA_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/linear -o lin -- python3 $R/bench.py --in-flight 1 --cpu-seconds 0 --no-single-image --no-standalone --no-cli --no-kernel-events --no-pmc > $OUT/bench_linear_under_rocprof.json 2> $OUT/linear.log &&
for cfg in "2048 6 0" "3072 0 0" "4096 8 1"; do set -- $cfg; rocprofv3 --kernel-trace --output-format csv -d $OUT/tl_$1 -o t -- python3 $R/devtools/single_image_trace.py $1 $2 $3 > $OUT/tl_$1.log 2>&1 || exit 1; done
cd $R && python devtools/trace_levels.py $OUT/linear/lin_kernel_trace.csv > $OUT/kernel_medians_linear_context_C4.txt &&
for n in 2048 3072 4096; do python devtools/step_timeline.py $OUT/tl_$n/t_kernel_trace.csv 14 > $OUT/timeline_single_$n.txt; done
