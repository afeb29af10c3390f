#!/bin/bash
# A/B of kernel forms on the GPU box (through gpurun): one lone linear context per configuration under the kernel tracer, per-kernel
# medians by devtools/trace_levels.py. Usage: bash devtools/ab_job.sh <tag> <N> <L> <batch> "NAME ENV=VAL ENV=VAL" "NAME2 ..." ...
set -o pipefail
TAG=$1; N=$2; L=$3; B=$4; shift 4
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    set -- $cfg
    name=$1; shift
    (
        for kv in "$@"; do export "$kv"; done
        rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/devtools/linear_probe.py $N $L $B > $OUT/$name.log 2>&1
    ) || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
    python3 $R/devtools/trace_levels.py $(find $OUT/$name -name "t_kernel_trace.csv" | head -1) > $OUT/$name.txt
    echo "== $name"; head -14 $OUT/$name.txt
    rm -rf $OUT/$name
done
