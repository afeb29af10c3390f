set -e
python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
for r in 1 2; do
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-standalone --no-single-image 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], {k:v['mean_us'] for k,v in d['kernels'].items()})"
done
TIF_LINEAR=1 TIF_MAX=3 python devtools/two_in_flight.py
