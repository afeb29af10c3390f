set -e
python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python bench.py > gpurun_out/bench_C4.json 2> gpurun_out/bench_C4.err
for w in C2 C3 C5; do python bench.py --workload $w --cpu-seconds 3 --no-standalone > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; done
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-standalone --no-single-image --no-kernel-events 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('driver form', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'])"
bash devtools/gpu_profile_job.sh r02e
