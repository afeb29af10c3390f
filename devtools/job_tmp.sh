set -e
python bench.py > gpurun_out/bench_C4.json 2> gpurun_out/bench_C4.err
for w in C2 C3 C5; do python bench.py --workload $w --cpu-seconds 3 --no-standalone > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; done
bash devtools/gpu_profile_job.sh r02b
