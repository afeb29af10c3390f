"""Dev tool: run bench.py under several env settings and print selected kernel times.
usage: sweep_env.py kernel1,kernel2 VAR=a,b VAR2=c,d ..."""
import os, sys, itertools, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernels = sys.argv[1].split(",")
vars_ = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[2:]]
for combo in itertools.product(*[v for _, v in vars_]):
    env = dict(os.environ)
    for (k, _), val in zip(vars_, combo):
        env[k] = val
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "2", "--cpu-seconds", "0"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        ks = " ".join("%s=%.1f" % (k, d["kernels"][k]["mean_us"]) for k in kernels)
        print(" ".join("%s=%s" % (k, v) for (k, _), v in zip(vars_, combo)), "| ms/step %.3f |" % d["ms_per_step"], ks, flush=True)
    except Exception as e:
        print(combo, "FAILED", r.stderr[-300:], flush=True)
