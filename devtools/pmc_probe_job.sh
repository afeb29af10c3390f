#!/bin/bash
# counters of one kernel of the lone linear context, for several library builds: bash devtools/pmc_probe_job.sh <tag> <kernel substring> "NAME ENV=VAL ..." ...
set -o pipefail
TAG=$1; PAT=$2; shift 2
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
    set -- $cfg; name=$1; shift
    (
        for kv in "$@"; do export "$kv"; done
        rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/$name -o t -- python3 $R/devtools/linear_probe.py 2048 6 8 > $OUT/$name.log 2>&1
    ) || { echo "$name failed"; tail -5 $OUT/$name.log; exit 1; }
    echo "== $name"; python3 $R/devtools/pmc_probe_summary.py $OUT/$name "$PAT" | tee $OUT/$name.txt
    rm -rf $OUT/$name
done
