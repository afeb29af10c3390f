#!/bin/bash
# A/B of two source TREES on the GPU box: bash devtools/ab_tree_job.sh <tag> <N> <L> <batch> "NAME TREE ENV=VAL ..." ...   (TREE = . or _r03)
set -o pipefail
TAG=$1; N=$2; L=$3; B=$4; shift 4
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    set -- $cfg; name=$1; tree=$2; shift 2
    ( for kv in "$@"; do export "$kv"; done
      rocprofv3 --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/$tree/devtools/linear_probe.py $N $L $B > $OUT/$name.log 2>&1 ) || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
    python3 $R/devtools/trace_levels.py $(find $OUT/$name -name "t_kernel_trace.csv" | head -1) > $OUT/$name.txt
    echo "== $name"; head -8 $OUT/$name.txt
    rm -rf $OUT/$name
done
