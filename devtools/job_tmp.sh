set -e
python -m pytest tests -q -x -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
echo "== LINEAR flag, up to 6" >> gpurun_out/tif4.log
TIF_LINEAR=1 TIF_MAX=6 python devtools/two_in_flight.py >> gpurun_out/tif4.log 2>&1
echo "== LINEAR flag, torch runtime" >> gpurun_out/tif4.log
TIF_TORCH=1 TIF_LINEAR=1 TIF_MAX=5 python devtools/two_in_flight.py >> gpurun_out/tif4.log 2>&1
echo "== LINEAR flag, single image" >> gpurun_out/tif4.log
TIF_LINEAR=1 TIF_MAX=5 python devtools/two_in_flight.py 2048 6 1 400 >> gpurun_out/tif4.log 2>&1
for a in "--in-flight 4" "--in-flight 1" "--in-flight 4" "--in-flight 2"; do
  echo "bench --steps 20 --warmup 5 $a" >> gpurun_out/tif4.log
  MUSICA_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 $a --cpu-seconds 0 --no-standalone --no-single-image --no-kernel-events 2>> gpurun_out/tif4.log | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'], d['one_context'])" >> gpurun_out/tif4.log
done
