set -e
python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for r in 1 2 3; do
MUSICA_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-standalone --no-single-image --no-kernel-events 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['ms_per_step'], d['value'], d['one_context']['ms_per_step'])
    else: print(l.strip())"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02c/stats1 -o bench1 -- python3 $GRAFT_REPO_ROOT/bench.py --in-flight 1 --cpu-seconds 0 --no-single-image --no-standalone > $GRAFT_REPO_ROOT/gpurun_out/r02c_bench1.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02c_stats1.log
