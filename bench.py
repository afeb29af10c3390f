#!/usr/bin/env python3
"""bench.py — full-pipeline MUSICA throughput on N MI355X GPUs + roofline of the metric kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C4|C2|C3|C5|REF]

One "step" = one pass of the whole hot path (minmax -> pyramid reduce -> analysis -> contrast /
noise-reduction + expand -> gradation) over one batch of synthetic raw images already resident in
HBM. Steps alternate over --in-flight contexts (default 3, the C ABI's musica_pipeline_*: step s is enqueued on context s mod 3, each
context with its own copy of the input, its own buffers and ONE in-order stream, so the chip-filling kernels of a step
run in the part-idle phases of the two steps beside it); `one_context` in the JSON line is the same K steps on a single
context with the library's default dispatch for that batch (its `what` names it), each step behind the previous one. After the W warm-up steps the whole job (the same K steps + its
tail) is rehearsed once, untimed; the timed region is then EXACTLY K steps between a barrier + device synchronize on both
sides. Default workload C4 is BASELINE.json configs[3] seen from one GPU:
8 independent 2048 x 2048 16-bit images, 6-level pyramid, per GPU and per step (weak scaling: at N = 8
that is the 64-image batch, at N = 1 it is 8 x configs[1]). Image k of the N x 8 images of a step goes to
rank k mod N (batch.assign_images) with no data-path collective; RCCL is used once, inside the
timed region, to all-gather the per-image summary statistics (musica_stats) of the final step.

`--gpus N` with N > 1 started plainly (no WORLD_SIZE in the environment) launches its own N ranks through
`python -m torch.distributed.run` before anything touches the GPU, relays rank 0's JSON line and fails if
the ranks do not all join; under an external launcher (WORLD_SIZE set) it is one rank of that job.

Prints ONE JSON line on rank 0 (contract in the task statement): value = megapixels/s of the whole job, plus
  roofline            : the metric kernel BASELINE.json names — fused 5-tap smooth + 2x downsample
                        (k_reduce_dma) on 4096 x 4096 f32 — timed with HIP events on the library's stream
                        over back-to-back launches that ROTATE over 8 distinct input / output planes (640 MB,
                        so no launch finds its data in the 256 MiB Infinity Cache): an HBM number;
                        copy_ceiling = a plain streaming kernel of the same traffic shape timed the same way;
                        at_8192 = the same measurement at 8192 x 8192 (3 plane pairs, 1 GB);
  roofline_4096_warm  : the same launches on ONE input plane (cache-resident: 80 MB inside the Infinity Cache);
  roofline_pipeline_l0: the level-0 launch of that kernel inside the pipeline (raw uint16 input normalised on
                        the fly, 3 B/px), HIP events around it in a pass over the same K steps;
  kernels             : per-kernel-family mean duration, algorithmic GB/s and share of the step;
  cpu_baseline        : the CPU oracle (a port: the reference has no CPU path) on a bounded sample;
  reference_3072_L12  : one 3072 x 3072 image per execute with the reference's 12 levels, one context (the reference's call shape);
  cli                 : wall time of `musica-standalone image.raw out.bmp` at that size, 5 fresh processes, and its own timing line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

WORKLOADS = {
    # BASELINE.md section 2 names: (image side, levels, images per GPU per step, bits, flags, description)
    "C4": (2048, 6, 8, 16, 0, "C4: 8 x 2048x2048 u16 per GPU per step, 6-level pyramid (BASELINE configs[3] shard = 8 x configs[1])"),
    "C2": (2048, 6, 1, 16, 0, "C2: 1 x 2048x2048 u16, 6-level pyramid (BASELINE configs[1])"),
    "C3": (4096, 8, 1, 16, 1, "C3: 1 x 4096x4096 u16, 8-level pyramid + CLAHE gradation (BASELINE configs[2])"),
    "C5": (8192, 10, 1, 12, 0, "C5: 1 x 8192x8192 12-bit, 10-level pyramid, noise reduction on (BASELINE configs[4])"),
    # the reference's own call: one 3072 x 3072 image (test/standalone/main.cpp:31), L = ceil(log2 N) = 12 (src/vk_processing.cpp:1989)
    "REF": (3072, 12, 1, 16, 0, "REF: 1 x 3072x3072 u16, 12-level pyramid: the configuration maverick-standalone runs (test/standalone/main.cpp:31)"),
}
COLD_BUFFERS = 8          # 8 x (64 MB in + 16 MB out) = 640 MB rotating footprint for the HBM measurement
COLD_ITERS = 64


def algorithmic_bytes(n, levels, batch, fused_u16=True, fused_gradhist=False, fused_rb=0, le090=False, fused_sdev=False):
    """Algorithmic HBM bytes per launch of each kernel family, f32 = 4 B, u16 = 2 B.
    fused_u16: the level-0 kernels read the raw uint16 pixels (2 B/px) instead of a stored normalized image (4 B/px);
    fused_rb: 1 = reduce + band of level 0 are one launch, 2 = of every level whose side is a multiple of 8 (then the
    `reduce_*` families carry the band image too and no `band_*` launch remains at those levels);
    le090: the level-0 reduce + band launch also writes the 1 bit/px `normalized <= 0.9` image the expand launch reads;
    fused_sdev: the expand launches of levels 0 .. 2 compute sdev from the band image themselves (musica_fuses_sdev): the sdev launches of
    those levels only read (4 B/px) and the expand launches read no sdev image."""
    src = 2 if fused_u16 else 4
    s = [n]
    for _ in range(levels):
        s.append((s[-1] + 1) // 2)
    p = [v * v for v in s]
    rest = range(1, levels)
    rb_rest = [i for i in rest if fused_rb >= 2 and s[i] % 8 == 0 and s[i] >= 8]
    mask = p[0] / 8.0 if le090 else 0.0
    raw_in_expand = src if (fused_gradhist and not le090) else 0
    return {
        "minmax": 2 * p[0] * batch,
        "normalize": 6 * p[0] * batch,
        # read S^2 once (u16 when fused), write (S/2)^2 f32 once (+ the band image and the <= 0.9 bits when reduce and band are one launch)
        "reduce_l0": ((src + 4) * p[0] + 4 * p[1] + mask) * batch if fused_rb else (src * p[0] + 4 * p[1]) * batch,
        "reduce_rest": sum((8 if i in rb_rest else 4) * p[i] + 4 * p[i + 1] for i in rest) * batch / max(1, len(rest)),
        "band_l0": ((src + 4) * p[0] + 4 * p[1]) * batch,                # read fine + coarse, write band
        "band_rest": sum(8 * p[i] + 4 * p[i + 1] for i in rest if i not in rb_rest) * batch / max(1, len(rest) - len(rb_rest)),
        "sdev_hist": sum((4 if (fused_sdev and i < 3) else 8) * p[i] for i in range(4)) * batch / 4.0,      # read band, write sdev (hist in LDS)
        "expand_l0": (((8 if fused_sdev else 12) + raw_in_expand) * p[0] + 4 * p[1] + (mask if fused_gradhist else 0)) * batch,   # read band (+ sdev) + coarse (+ raw or bits), write recon
        "expand_rest": (sum((8 if (fused_sdev and i < 3) else 12) * p[i] + 4 * p[i + 1] for i in range(1, 4)) +
                        sum(8 * p[i] + 4 * p[i + 1] for i in range(4, levels))) * batch / max(1, len(rest)),
        "grad_hist": 0 if fused_gradhist else (4 + src) * p[0] * batch,  # read recon + normalized (or raw); fused: the launch only recounts images with exact zeros
        "grad_apply": 8 * p[0] * batch,                                  # read recon, write graded
        "curves": 4 * 2048 * 4 * batch,
        "cnr": 8 * p[3] * batch,
        "grad_curve": 4 * 1024 * batch,
    }


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C4", type=str.upper, choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="time budget of the CPU baseline sample (0 = skip)")
    ap.add_argument("--in-flight", type=int, default=3,
                    help="contexts whose steps alternate (musica_pipeline_*): step s runs on context s mod D, so the head of a step fills the "
                         "part-idle tail of the one before; 1 = one context, every step behind the previous one")
    ap.add_argument("--batch", type=int, default=0, help="override the workload's images per GPU per step (experiments only)")
    ap.add_argument("--kernel-events", action="store_true",
                    help="bracket the level-0 metric kernel with HIP events inside the timed steps (forces eager launches: stream "
                         "capture drops event records, so the default timed region replays the hipGraph without events)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo + MUSICA_BENCH_ONE_DEVICE=1 rehearses the multi-rank flow on a one-GPU box (every rank on cuda:0, "
                         "the stats rows travel through host memory)")
    ap.add_argument("--no-single-image", action="store_true", help="skip the one-image-per-execute measurement (keeps a kernel trace of the run pure)")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the per-kernel event passes after the timed region")
    ap.add_argument("--no-standalone", action="store_true", help="skip the stand-alone 4096^2 metric-kernel measurements (roofline becomes the in-pipeline launch)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the metric kernel's HBM traffic with rocprofv3 --pmc (roofline.traffic then comes from profiles/pmc_traffic.json)")
    ap.add_argument("--no-cli", action="store_true", help="skip the drop-in measurement (musica-standalone <raw> <bmp> in fresh processes at 3072^2 / L12)")
    return ap.parse_args()


def self_launch(args):
    """`bench.py --gpus N` started without a launcher: become the launcher. Nothing in this process has touched
    torch or HIP yet (a process that has initialised the GPU must not be replaced or forked into ranks)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MUSICA_BENCH_CHILD="1")
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if r.returncode != 0 or line is None:
        raise SystemExit("bench.py --gpus %d: the rank launcher exited with code %d%s" % (args.gpus, r.returncode, "" if line else " and printed no result line"))
    got = json.loads(line)
    if got.get("n_gpus") != args.gpus:
        raise SystemExit("bench.py --gpus %d: the result line reports n_gpus = %r" % (args.gpus, got.get("n_gpus")))
    print(line)


def measure_metric_traffic(timeout=90):
    """HBM-side bytes per launch of the stand-alone metric kernel at 4096^2 from HBM, collected NOW: two rocprofv3 passes
    (FETCH_SIZE and WRITE_SIZE cannot share a pass, MI355X_MICROARCH.md 'rocprofv3 PMC slots') of devtools/pmc_metric_target.py as child
    processes, --pmc with no trace option. Units and corrections as that guide prescribes for gfx950: both counters are KiB; FETCH_SIZE
    tallies wide streaming reads at half their size (x 2); WRITE_SIZE is exact. Returns (bytes, detail) or (None, reason)."""
    import csv
    import glob
    import shutil
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    # never start a profiler from under a profiler: the child would inherit the outer tool's preload environment, and a launcher that
    # execs its target with a profiler library preloaded is the exec-after-GPU-init pattern this pool forbids (round-3 advisory)
    if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "already under a profiler"
    target = os.path.join(ROOT, "devtools", "pmc_metric_target.py")
    got = {}
    td = tempfile.mkdtemp(prefix="musica_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(td, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--", sys.executable, target]
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=td, env=dict(os.environ, TMPDIR=td))
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out" % counter
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s failed (exit %d)" % (counter, r.returncode)
            rows = []
            for row in csv.DictReader(open(files[0], newline="")):
                if row["Counter_Name"] == counter and "k_reduce_dma<4>" in row["Kernel_Name"]:
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            rows.sort()
            vals = [v for _, v in rows][-16:]          # the 16 counted launches (the first rotation is the warm-up)
            if len(vals) < 16:
                return None, "only %d k_reduce_dma<4> dispatches in the %s pass" % (len(vals), counter)
            got[counter] = sum(vals) / len(vals)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    fetch, write = 2.0 * 1024.0 * got["FETCH_SIZE"], 1024.0 * got["WRITE_SIZE"]
    return int(round(fetch + write)), {"fetch_bytes": int(round(fetch)), "write_bytes": int(round(write)), "launches_averaged": 16}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            return self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): start %d ranks, or run `python bench.py --gpus %d` without a launcher" % (world, args.gpus, args.gpus, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if os.environ.get("MUSICA_BENCH_ONE_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch  # first: libmusica_hip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import batch as mb
    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd import processing as mp
    from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import phantom

    if not torch.cuda.is_available() or mp.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the MUSICA path has no CPU fallback")
    if torch.cuda.device_count() < local_rank + 1:
        raise SystemExit("rank %d wants cuda:%d but only %d device(s) are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus asked for %d" % (dist.get_world_size(), args.gpus))

    n, levels, batch, bits, flags, desc = WORKLOADS[args.workload]
    if args.batch > 0:
        batch = args.batch
        desc += " [--batch %d override]" % batch
    image_ids = mb.assign_images(world * batch, world)[rank]      # image k -> rank k mod world
    assert len(image_ids) == batch
    px = np.stack([phantom(n, 100 + k, bits=bits) for k in image_ids])   # C4 uses default_rng(100 + k)

    depth = 1 if args.kernel_events else max(1, args.in_flight)   # events inside the timed steps need the one profiled context
    try:
        pipe = mp.MusicaPipeline(n, levels=levels, batch=batch, depth=depth, flags=flags, device=local_rank)   # the C ABI's musica_pipeline_*
    except RuntimeError as e:
        raise SystemExit(str(e))
    pipe.upload(px)                                                # inputs resident in HBM (one copy per context) before the timed region
    pipe.prime()                                                   # every context has captured its graph; the best set of hardware queues is kept
    queue_calibration = {str(k): round(v, 4) for k, v in pipe.calibration().items()} or None
    sdev_in_expand = bool(pipe.context(0).fuses_sdev())              # the timed contexts compute sdev inside the expand launches of levels 0 .. 2 (no stored sdev images there)
    if depth == 1:
        proc = pipe.context(0)
    else:                                                          # the per-kernel passes and the other measurements: one context alone, default dispatch
        proc = mp.MusicaProcessing(device=local_rank)
        if not proc.init(n, levels=levels, batch=batch, flags=flags):
            raise SystemExit("musica_create failed: " + mp.last_error())
        proc.upload(px)
        for _ in range(2):
            proc.execute_device()
        proc.sync()

    d_stats = torch.zeros((batch, mb.STATS_WORDS), dtype=torch.int32, device="cuda")

    def barrier():
        if distributed:
            dist.barrier()

    def step():
        try:
            pipe.step()
        except RuntimeError as e:
            raise SystemExit(str(e))

    def step_one():
        if not proc.execute_device():
            raise SystemExit("musica_execute_device failed: " + mp.last_error())

    def finish_job():
        """End of a job: per-image stats rows of the last step on the device (image ids job-wide: rank + index * world,
        batch.assign_images), every context drained, one all-gather."""
        pipe.last().stats_device(d_stats.data_ptr(), image_id_base=rank, image_id_stride=world)
        pipe.sync()
        return mb.gather_rows(d_stats, world, dist if distributed else None)   # RCCL over xGMI: the only inter-GPU traffic

    for _ in range(args.warmup):
        step()
    finish_job()                                                   # untimed rehearsal: loads torch's / RCCL's kernels, connects the ranks

    # The timed region replays the captured hipGraph (the product's default dispatch). ROCm 7.2 stream capture
    # drops hipEventRecord calls (devtools/graph_events.hip), so HIP events around a kernel need eager launches:
    # the level-0 metric kernel is bracketed in a second pass over the same K steps right after the timed region
    # (2 records per step), and all launches in a third pass (costs ~18 % of the step) for the "kernels" table.
    # --kernel-events moves the level-0 events into the timed region itself (eager launches, ~4 % slower).
    kernel_events = args.kernel_events
    proc.profile_reset()
    proc.profile_enable(["reduce_l0"] if kernel_events else False)

    # One untimed rehearsal of the whole job (the same K steps + the tail) right before the timed one: the first K-step run
    # after start-up is 3 - 5 % slower than every later one (7.86 ms against 7.52 - 7.59 for 20 steps, same process: clocks
    # and caches settle over more than the W = 5 warm-up steps the driver asks for). MUSICA_BENCH_REHEARSALS=0 turns it off.
    rehearsals = int(os.environ.get("MUSICA_BENCH_REHEARSALS", "1"))
    for rep in range(rehearsals):
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for _ in range(args.steps):
            step()
        tq = time.perf_counter()
        finish_job()
        torch.cuda.synchronize()
        if os.environ.get("MUSICA_BENCH_TRACE") == "1":
            sys.stderr.write("rehearsal %d: %d steps enqueued %.3f ms, job done %.3f ms\n" % (rep, args.steps, (tq - tr) * 1e3, (time.perf_counter() - tr) * 1e3))
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter()
    gathered = finish_job()
    t_fin = time.perf_counter()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    if os.environ.get("MUSICA_BENCH_TRACE") == "1":
        sys.stderr.write("timed region: %d steps enqueued after %.3f ms, finish_job returned after %.3f ms, end %.3f ms\n"
                         % (args.steps, (t_enq - t0) * 1e3, (t_fin - t0) * 1e3, (t1 - t0) * 1e3))

    elapsed = t1 - t0
    if distributed:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    st = gathered.cpu().numpy()
    got_ids = sorted(int(r[0]) for r in st)
    if st.shape[0] != world * batch or got_ids != list(range(world * batch)):
        raise SystemExit("stats gather incomplete: %d rows, ids %s (expected %d images from %d ranks)" % (st.shape[0], got_ids[:8], world * batch, world))

    proc.profile_enable(False)
    prof_timed = proc.profile()
    # the same K steps on ONE context (every step behind the previous one): reported beside `value`
    if depth > 1:
        pipe.cleanup()                                             # the timed pipeline's contexts are no longer needed
    step_one()
    proc.sync()
    ts0 = time.perf_counter()
    for _ in range(args.steps):
        step_one()
    proc.sync()
    one_ctx_ms = (time.perf_counter() - ts0) / args.steps * 1e3
    step = step_one
    prof = {}
    if not args.no_kernel_events:
        if not kernel_events:
            proc.profile_reset()
            proc.profile_enable(["reduce_l0"])
            for _ in range(args.steps):
                step()
            proc.sync()
            proc.profile_enable(False)
            prof_timed = proc.profile()
        # untimed pass over the same steps with every kernel family bracketed -> "kernels" table
        proc.profile_reset()
        proc.profile_enable(True)
        for _ in range(args.steps):
            step()
        proc.sync()
        proc.profile_enable(False)
        prof = proc.profile()
    if prof_timed.get("reduce_l0", (0, 0))[1]:
        prof["reduce_l0"] = prof_timed["reduce_l0"]              # the metric-kernel-only measurement wins

    result = None
    if rank == 0:
        mpix = world * batch * n * n * args.steps / 1e6
        ms_per_step = elapsed / args.steps * 1e3
        fused = n % 8 == 0
        rb_mode = 2 if proc.fuses_reduce_band() else 0
        le090 = bool(proc.fuses_gradhist() and rb_mode >= 1)
        ab = algorithmic_bytes(n, levels, batch, fused, fused_gradhist=proc.fuses_gradhist(), fused_rb=rb_mode, le090=le090, fused_sdev=proc.fuses_sdev())
        kernels = {}
        total_kernel_us = 0.0
        for name, (us, cnt) in prof.items():
            if cnt:
                total_kernel_us += us * cnt / args.steps
        for name, (us, cnt) in prof.items():
            if not cnt:
                continue
            gbs = ab.get(name, 0) / (us * 1e-6) / 1e9 if us > 0 else 0.0
            kernels[name] = {"mean_us": round(us, 2), "launches_per_step": cnt // args.steps, "alg_GBps": round(gbs, 1),
                             "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                             "share_of_step": round(us * cnt / args.steps / max(total_kernel_us, 1e-9), 4)}
        traffic_doc = {}
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic_doc = json.load(open(tpath))
            except Exception:
                traffic_doc = {}
        traffic_source = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of devtools/pmc_target.py, committed; not collected by this run)"
        pipeline_l0 = None
        if "reduce_l0" in kernels:
            k = kernels["reduce_l0"]
            pipeline_l0 = {"kernel": "%s (5-tap smooth + 2x downsample, level 0 of the pipeline, %d images of %dx%d per launch; input read as %s)"
                                     % ("k_reduce_band_u16 — smooth + downsample AND the band-pass image in one march" if proc.fuses_reduce_band() else
                                        "k_reduce_u16_pf" if fused else "k_reduce_dma", batch, n, n,
                                        "raw uint16 normalised on the fly: 2 B/px in + 1 B/px (+ 4 B/px band) out" if fused else "f32: 4 B/px in + 1 B/px out"),
                           "bound": "hbm", "achieved": k["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": round(k["alg_GBps"] / HBM_PEAK_GBS, 4),
                           "traffic": traffic_doc.get(args.workload, {}).get("reduce_l0_hbm_bytes_per_launch"), "traffic_source": traffic_source,
                           "algorithmic_bytes_per_launch": ab["reduce_l0"], "mean_us": k["mean_us"],
                           "measured": "HIP event pair around the launch on the library's stream, %d steps (eager launches)" % args.steps}
        roofline, warm = pipeline_l0, None
        if not args.no_standalone:
            # the BASELINE target: the metric kernel alone on 4096 x 4096 f32 images
            b4096 = 5 * 4096 * 4096
            cold_us, copy_us = proc.k_reduce_cold(4096, nbuf=COLD_BUFFERS, iters=COLD_ITERS)
            warm_us = proc.k_reduce_timed(4096, batch=1, iters=200)
            gb = lambda us: round(b4096 / (us * 1e-6) / 1e9, 1)
            roofline = {"kernel": "k_reduce_dma (fused 5-tap smooth + 2x downsample, LDS-DMA tiles) stand-alone on 4096x4096 f32, %d back-to-back launches rotating over "
                                  "%d distinct input / output planes (%d MB footprint > 256 MiB Infinity Cache: every launch reads from HBM)"
                                  % (COLD_ITERS, COLD_BUFFERS, COLD_BUFFERS * b4096 // 1000000),
                        "bound": "hbm", "achieved": gb(cold_us), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gb(cold_us) / HBM_PEAK_GBS, 4),
                        "traffic": traffic_doc.get("standalone_4096_cold_hbm_bytes_per_launch"), "traffic_source": traffic_source,
                        "algorithmic_bytes_per_launch": b4096, "mean_us": round(cold_us, 2),
                        "traffic_note": "FETCH_SIZE x 2 as the guide prescribes for 16-byte streaming reads; the excess over the algorithmic bytes is the same for every tile height "
                                        "(not row re-reads): the 4-byte halo-column gathers of the tiles, a width the guide marks as uncalibrated for the x 2 (DESIGN.md section 4)",
                        "measured": "one HIP event pair around the %d launches on the library's stream, right after the timed steps" % COLD_ITERS,
                        "copy_ceiling": {"kernel": "k_copy41: plain streaming kernel, same traffic shape (read S^2 f32, write (S/2)^2 f32), same rotation",
                                         "achieved": gb(copy_us), "unit": "GB/s", "mean_us": round(copy_us, 2),
                                         "metric_kernel_vs_ceiling": round(copy_us / cold_us, 4)}}
            # the same kernel at 8192 x 8192 (3 plane pairs = 1 GB rotating footprint), reported beside the 4096^2 figure
            b8192 = 5 * 8192 * 8192
            cold8_us, copy8_us = proc.k_reduce_cold(8192, nbuf=3, iters=24)
            roofline["at_8192"] = {"achieved": round(b8192 / (cold8_us * 1e-6) / 1e9, 1), "unit": "GB/s", "frac": round(b8192 / (cold8_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                   "mean_us": round(cold8_us, 2), "algorithmic_bytes_per_launch": b8192,
                                   "copy_ceiling_us": round(copy8_us, 2), "measured": "24 back-to-back launches rotating over 3 distinct 8192x8192 plane pairs (1 GB)"}
            if world == 1 and not args.no_pmc:
                tb, detail = measure_metric_traffic()
                if tb is not None:
                    roofline["traffic"] = tb
                    roofline["traffic_source"] = ("measured by this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no trace option) of devtools/pmc_metric_target.py "
                                                  "as child processes; KiB -> bytes, FETCH_SIZE x 2, WRITE_SIZE exact (MI355X_MICROARCH.md); mean of 16 launches of k_reduce_dma<4>")
                    roofline["traffic_detail"] = detail
                else:
                    roofline["traffic_source"] = traffic_source + " [live collection unavailable: %s]" % detail
            warm = {"kernel": "the same kernel, 200 back-to-back launches on ONE 4096x4096 input (80 MB: cache-resident in the 256 MiB Infinity Cache, not an HBM number)",
                    "bound": "infinity-cache", "achieved": gb(warm_us), "unit": "GB/s", "frac_of_hbm_peak": round(gb(warm_us) / HBM_PEAK_GBS, 4),
                    "mean_us": round(warm_us, 2), "algorithmic_bytes_per_launch": b4096}
        # PCIe-inclusive rate of the reference-shaped call (host pixels in, synchronous): never `value`
        te0 = time.perf_counter()
        for _ in range(3):
            if not proc.execute(px):
                raise SystemExit("musica_execute failed: " + mp.last_error())
        e2e = 3 * batch * n * n / 1e6 / (time.perf_counter() - te0)
        # ... and of the overlapped host path: a stream of batches through pinned staging, H2D of batch j+1 under compute of j
        e2e_stream = None
        reps = 8
        pinned = [proc.host_alloc(px.shape) for _ in range(2)]    # page-locked inputs: the H2D copies run at the PCIe rate
        for b in pinned:
            b[...] = px
        if not proc.execute_stream([pinned[j & 1] for j in range(2)]):   # allocates the second device buffer, captures its graph
            raise SystemExit("musica_execute_stream failed: " + mp.last_error())
        te0 = time.perf_counter()
        if not proc.execute_stream([pinned[j & 1] for j in range(reps)]):
            raise SystemExit("musica_execute_stream failed: " + mp.last_error())
        e2e_stream = reps * batch * n * n / 1e6 / (time.perf_counter() - te0)
        # BASELINE configs[3] as it is stated — ONE pass over a host-resident shard: pinned host pixels -> H2D -> one step -> stats rows,
        # wall time of musica_execute (image k's chain starts when image k has landed) + the stats read-back; median of 7 passes
        one_pass = []
        rows_dev = torch.zeros((batch, mb.STATS_WORDS), dtype=torch.int32, device="cuda:%d" % local_rank)
        def pass_once():
            tp0 = time.perf_counter()
            if not proc.execute(pinned[0]):
                raise SystemExit("musica_execute failed: " + mp.last_error())
            proc.stats_device(rows_dev.data_ptr(), 0, 1)
            proc.sync()
            rows_dev.cpu()
            return time.perf_counter() - tp0
        for _ in range(8):
            one_pass.append(pass_once())
        one_pass = sorted(one_pass[1:])
        one_pass_ms = one_pass[len(one_pass) // 2] * 1e3
        whole = None
        if batch > 1:   # the same pass with the batch copied and computed as a whole (MUSICA_HOST_LANES=0): what the lanes buy
            os.environ["MUSICA_HOST_LANES"] = "0"
            tw = []
            for _ in range(6):
                tw.append(pass_once())
            del os.environ["MUSICA_HOST_LANES"]
            tw = sorted(tw[1:])
            whole = tw[len(tw) // 2] * 1e3
        for b in pinned:
            proc.host_free(b)
        # BASELINE configs[1] beside the batched workload: ONE image of the same size per execute, as the reference's
        # VulkanProcessing::execute is called (a latency-bound chain of dependent kernels; reported, never `value`)
        single = None
        if batch > 1 and world == 1 and not args.no_single_image:
            p1 = mp.MusicaProcessing(device=local_rank)
            if p1.init(n, levels=levels, batch=1, flags=flags):
                p1.upload(px[:1])
                for _ in range(max(args.warmup, 3)):
                    p1.execute_device()
                p1.sync()
                ts0 = time.perf_counter()
                for _ in range(args.steps):
                    p1.execute_device()
                p1.sync()
                ts = (time.perf_counter() - ts0) / args.steps
                single = {"workload": "1 x %dx%d per execute, %d-level pyramid (BASELINE configs[1] shape)" % (n, n, levels),
                          "value": round(n * n / 1e6 / ts, 1), "unit": "MP/s", "ms_per_image": round(ts * 1e3, 4), "dispatch": p1.dispatch_text()}
                p1.cleanup()
        # The reference's own call shape and the drop-in itself: one 3072 x 3072 image, L = 12 (test/standalone/main.cpp:31,
        # src/vk_processing.cpp:1989). (a) one context, one image per execute, resident input; (b) `musica-standalone <raw> <bmp>` as
        # test/metamorphic_test/script.py:200-214 spawns it: wall time of fresh processes (median of 5) and the CLI's own timing line.
        ref = None
        cli = None
        if world == 1 and not args.no_cli:
            import re
            import tempfile
            rn = 3072
            rpx = phantom(rn, 31)
            pr = mp.MusicaProcessing(device=local_rank)
            if pr.init(rn, levels=0, batch=1, flags=0):
                pr.upload(rpx[None])
                for _ in range(3):
                    pr.execute_device()
                pr.sync()
                tr0 = time.perf_counter()
                for _ in range(args.steps):
                    pr.execute_device()
                pr.sync()
                tr_dev = (time.perf_counter() - tr0) / args.steps
                tr0 = time.perf_counter()
                for _ in range(3):
                    pr.execute(rpx)
                tr_host = (time.perf_counter() - tr0) / 3
                ref = {"workload": "1 x 3072x3072 per execute, 12-level pyramid (the reference's configuration), one context",
                       "ms_per_image_resident_input": round(tr_dev * 1e3, 4), "MPps_resident_input": round(rn * rn / 1e6 / tr_dev, 1),
                       "ms_per_image_host_input_synchronous": round(tr_host * 1e3, 4), "dispatch": pr.dispatch_text()}
                pr.cleanup()
            with tempfile.TemporaryDirectory() as td:
                from metamorphic_testing_of_the_musica_algorithm_for_x_ray_image_processing_amd.phantom import write_raw
                raw, out = os.path.join(td, "image.raw"), os.path.join(td, "out.bmp")
                write_raw(raw, rpx)
                walls, lines = [], []
                for _ in range(5):
                    tc0 = time.perf_counter()
                    r = subprocess.run([mp.CLI_PATH, raw, out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                    walls.append((time.perf_counter() - tc0) * 1e3)
                    if r.returncode != 0:
                        raise SystemExit("musica-standalone failed: " + r.stderr)
                    m = re.search(r"init: ([0-9.]+) .*exec: ([0-9.]+) .*save: ([0-9.]+) .*tot: ([0-9.]+)", r.stdout)
                    m2 = re.search(r"hip start-up: ([0-9.]+) .*create: ([0-9.]+) .*read: ([0-9.]+)", r.stdout)
                    m3 = re.search(r"cleanup: ([0-9.]+)", r.stdout)
                    if m:
                        lines.append([float(v) for v in m.groups()] + ([float(v) for v in m2.groups()] if m2 else [0.0, 0.0, 0.0]) + [float(m3.group(1)) if m3 else 0.0])
                order = sorted(range(len(walls)), key=lambda i: walls[i])
                mid = order[len(order) // 2]
                cli = {"command": "musica-standalone image.raw out.bmp (3072x3072, L = 12, no flags), 5 fresh processes",
                       "wall_ms_median": round(walls[mid], 1), "wall_ms_all": [round(w, 1) for w in walls],
                       "own_line_ms_of_the_median_run": dict(zip(["init", "exec", "save", "tot", "hip_startup", "create", "read", "cleanup"], lines[mid])) if len(lines) == len(walls) else None,
                       "note": "init = hip_startup (first HIP call) + create (device allocation, streams) + read (raw file); exec = H2D + pipeline incl. the first-launch "
                               "code-object loads; save = page-locked file image (5 ms), the BMP's 24-bpp rows written by the device straight into it, one 28 MB write (MUSICA_TIMING=1 prints the phases); cleanup = musica_destroy; "
                               "wall - tot - cleanup = process start (dynamic loading of the HIP runtime) and exit. No autotune, no graph capture in one-shot use."}
        # CPU baseline: the oracle (a port — the reference has no CPU path), all host cores, bounded sample
        cpu = None
        if args.cpu_seconds > 0 and world == 1:
            from oracle import binding as ob
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16))                         # a one-GPU box owns a 16-core share of the host
            ob.set_threads(cores)
            o = ob.Oracle(n, levels, ob.ORDER_FAST, flags & 1)
            o.execute(px[0])                                       # warm-up (page faults, thread pool)
            done, tc0 = 0, time.perf_counter()
            while True:
                o.execute(px[done % batch])
                done += 1
                if time.perf_counter() - tc0 >= args.cpu_seconds or done >= 4096:
                    break
            tc = time.perf_counter() - tc0
            # BASELINE configs[0] as well: one 512 x 512 image, 4 levels, literal 25-tap order, one thread (the semantic baseline)
            ob.set_threads(1)
            p0 = phantom(512, 1)
            o0 = ob.Oracle(512, 4, ob.ORDER_REFERENCE, 0)
            o0.execute(p0)
            n0, t00 = 0, time.perf_counter()
            while n0 < 64 and (n0 == 0 or time.perf_counter() - t00 < 2.0):
                o0.execute(p0)
                n0 += 1
            t0c = time.perf_counter() - t00
            ob.set_threads(cores)
            configs0 = {"value": round(n0 * 512 * 512 / 1e6 / t0c, 2), "unit": "MP/s", "cores": 1,
                        "sample": "%d x 512x512 images, 4-level pyramid, oracle MUSICA_ORDER_REFERENCE (literal 25-tap stencils), 1 thread, %.1f s" % (n0, t0c)}
            cpu = {"value": round(done * n * n / 1e6 / tc, 2), "unit": "MP/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
                   "host_cpus": os.cpu_count(), "configs0_single_thread": configs0,
                   "sample": "%d x %dx%d images, %d-level pyramid, oracle MUSICA_ORDER_FAST with OpenMP on %d threads, %.1f s"
                             % (done, n, n, levels, cores, tc)}
        result = {
            "metric": "megapixels/sec full MUSICA pipeline", "value": round(mpix / elapsed, 1), "unit": "MP/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_is": "whole-job MP/s of %d steps enqueued round-robin on %d one-stream contexts of each GPU (steps in flight), every context replaying its own HBM-resident copy "
                        "of the rank's %d-image shard (inputs do not change between steps; H2D excluded); `one_context`, `single_image`, `reference_3072_L12`, `e2e_host_*` "
                        "and `cli` are the same pipeline in the reference's one-frame-at-a-time call shapes" % (args.steps, depth, batch),
            "config": {"workload": desc, "image_size": n, "levels": levels, "images_per_gpu_per_step": batch, "contexts_in_flight": depth, "untimed_rehearsals_of_the_job": rehearsals,
                       "queue_calibration_ms": queue_calibration,
                       "sdev_in_expand_launches": sdev_in_expand,
                       "input": "seeded phantoms, %d-bit" % bits, "dispatch": "eager launches" if (kernel_events or os.environ.get("MUSICA_GRAPH", "1") == "0") else "hipGraph replay",
                       "kernel_events_in_timed_region": kernel_events, "sharding": "image k -> rank k mod N, no data-path collective",
                       "stats_gathered": int(st.shape[0]), "ranks_joined": world,
                       "collective": ("%s all_gather_into_tensor of %d stats rows per rank" % ("RCCL (torch backend nccl)" if args.backend == "nccl" else "gloo", batch)) if distributed else "none (one rank)"},
            "parity": "timed path = default (separable) order: bit-identical to the build's CPU oracle in that order; with MUSICA_FLAG_REFERENCE_ORDER the GPU runs the "
                      "shaders' literal 25-tap order, bit-identical to the oracle's literal restatement on every BASELINE config and 3072/L12 (tests/test_gpu_reference_order.py). "
                      "Default vs literal order, measured at full size on every config (profiles/r03_literal_order_*.json): stencil outputs within 6e-7 / 1e-6, every "
                      "noise-histogram argmax equal, reconstruction within 4e-6 on all but <= 0.003 % of the texels (max 2.1e-2: blocks under cnr texels at the "
                      "noise-reduction thresholds 3 / 9), <= 0.011 % of the 8-bit pixels differ (isolated 0 <-> 255 flips at getY's x > 1 -> 0 edge). "
                      "Parity with the reference itself is unpinned (no vectors in the reference, GLSL not buildable here)",
            "roofline": roofline, "roofline_4096_warm": warm, "roofline_pipeline_l0": pipeline_l0, "cpu_baseline": cpu, "kernels": kernels,
            "one_context": {"ms_per_step": round(one_ctx_ms, 4), "value": round(batch * n * n / 1e6 / (one_ctx_ms * 1e-3), 1), "unit": "MP/s per GPU",
                            "what": "the same %d steps on one context alone (%s), each step behind the previous one (rank 0)" % (args.steps, proc.dispatch_text())},
            "e2e_host_MPps": round(e2e, 1),
            "job_one_pass": {"what": "BASELINE configs[3] per GPU as stated: ONE pass over a host-resident shard — %d x %dx%d pinned host pixels -> H2D -> one step -> "
                                     "stats rows (musica_execute: image k's chain starts when image k has landed; median of 7 passes; PCIe-inclusive, never `value`)" % (batch, n, n),
                             "wall_ms": round(one_pass_ms, 4), "value": round(batch * n * n / 1e6 / (one_pass_ms * 1e-3), 1), "unit": "MP/s",
                             "pcie_bound_ms": round(batch * n * n * 2 / 63e9 * 1e3, 4), "fraction_of_pcie_bound": round(batch * n * n * 2 / 63e9 * 1e3 / one_pass_ms, 3),
                             "whole_batch_wall_ms": None if whole is None else round(whole, 4)},
            "e2e_host_overlapped": {"value": round(e2e_stream, 1), "unit": "MP/s", "what": "musica_execute_stream over %d batches in pinned host memory: H2D of batch j+1 under the kernels of batch j (PCIe-inclusive; never `value`)" % reps,
                                    "pcie_bound_MPps": round(63e9 / 2 / 1e6, 1), "fraction_of_device_rate": round(e2e_stream / (mpix / elapsed), 3)},
            "single_image": single,
            "reference_3072_L12": ref,
            "cli": cli,
        }
    if depth > 1:
        proc.cleanup()
    pipe.cleanup()
    if distributed:
        dist.barrier()          # rank 0 measured the stand-alone kernel after the timed region: leave together
        dist.destroy_process_group()
    if result is not None:
        print(json.dumps(result))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
