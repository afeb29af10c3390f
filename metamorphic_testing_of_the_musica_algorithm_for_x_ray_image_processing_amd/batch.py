"""Batch driver: independent images sharded one-shard-per-GPU, statistics gathered with one collective.

The reference processes one image per process on one device. A batch of independent images shards
trivially: rank r owns the images `assign_images(total, world)[r]`, runs them through its
own `MusicaProcessing` context with no data-path communication, and at the end one all-gather (RCCL over
xGMI on GPUs, gloo in the CPU tests) collects the fixed-size `musica_stats` row of every image.
"""
import ctypes as C

import numpy as np

from .processing import FLAG_LINEAR, MusicaProcessing, Stats, last_error

STATS_WORDS = C.sizeof(Stats) // 4  # 17 x 32-bit words per image


def assign_images(total, world):
    """Image k goes to rank k mod world; returns the list of image ids of every rank."""
    return [list(range(r, total, world)) for r in range(world)]


def stats_to_row(st):
    """One `musica_stats` (ctypes) -> int32[STATS_WORDS] with the struct's exact bytes."""
    return np.frombuffer(bytes(st), dtype=np.int32).copy()


def row_to_stats(row):
    return Stats.from_buffer_copy(np.ascontiguousarray(row, dtype=np.int32).tobytes())


def gather_rows(local_rows, world, dist=None, device=None):
    """All-gather equal-sized int32 row blocks; returns int32[world * rows, STATS_WORDS] on every rank.

    `local_rows` is a numpy int32 array (CPU / gloo) or a torch tensor already on `device` (GPU / RCCL).
    """
    import torch
    if isinstance(local_rows, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_rows, dtype=np.int32))
        if device is not None:
            t = t.to(device)
    else:
        t = local_rows
    if world == 1 or dist is None:
        return t                 # one rank: the rows are the job's rows (no copy, no kernel)
    if t.is_cuda and dist.get_backend() == "gloo":
        # one-GPU rehearsal of the multi-rank flow (bench.py --backend gloo): the rows travel through host memory
        parts = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]
        dist.all_gather(parts, t.cpu().contiguous())
        return torch.cat(parts).to(t.device)
    out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t.contiguous())
    return out


def summarize(rows):
    """Per-image dicts from gathered rows, sorted by image id."""
    out = []
    for r in np.asarray(rows.cpu() if hasattr(rows, "cpu") else rows):
        s = row_to_stats(r)
        out.append({"image_id": int(s.image_id), "min_sqrt": s.min_sqrt, "max_sqrt": s.max_sqrt,
                    "noise_max_bin": [int(v) for v in s.noise_max_bin], "grad_max_bin": int(s.grad_max_bin),
                    "mean_cnr": s.mean_cnr, "t0": s.t0, "ta": s.ta, "t1": s.t1})
    return sorted(out, key=lambda d: d["image_id"])


def process_shard(proc, images, image_ids):
    """Run the images of one rank (batch = proc.batch per execute) and return their stats rows as int32."""
    b = proc.batch
    n = len(image_ids)
    rows = np.zeros((n, STATS_WORDS), dtype=np.int32)
    for start in range(0, n, b):
        chunk = images[start:start + b]
        if len(chunk) < b:  # pad the last chunk by repeating its final image; padded rows are dropped
            chunk = np.concatenate([chunk, np.repeat(chunk[-1:], b - len(chunk), axis=0)])
        if not proc.execute(chunk):
            raise RuntimeError("musica_execute failed")
        for k in range(min(b, n - start)):
            st = proc.stats(k)
            st.image_id = image_ids[start + k]
            rows[start + k] = stats_to_row(st)
    return rows


class ShardPipeline:
    """`depth` contexts of ONE GPU whose steps alternate: step s is enqueued on context s mod depth.

    A step is a chain of dependent launches: chip-filling kernels at level 0, then small-level kernels, curve kernels and
    the gaps between dependent launches, during which a lone context leaves most of the GPU idle (~a sixth of a step).
    Contexts in flight fill each other's bubbles. Each context here is created with MUSICA_FLAG_LINEAR: ONE in-order
    stream, so consecutive contexts land on different hardware queues (the runtime has 4) and no step waits for an
    event of another queue. 8 x 2048^2 / L6 on MI355X, the build that introduced this: one three-stream context 0.484 ms per
    step, one linear context 0.50, two / three / four / six linear contexts in flight 0.41 / 0.375 / 0.405 / 0.375 ms (end
    of round 2: 0.44 / 0.46 / 0.38 / 0.35 - 0.36; DESIGN.md, "Steps in flight"; one 2048^2 image per step: 0.213 -> 0.093 ms;
    more than 4 hardware queues are worse).
    Every context owns its buffers, stream and captured graph; a step's results are bit-identical to a lone
    context's (tests/test_gpu_parity.py). The reference has one VulkanProcessing per process and one frame in flight
    (src/vk_processing.cpp:2104-2601); this is the batch driver's throughput form of it. The C ABI has the same object
    (musica_pipeline_*, processing.MusicaPipeline — what bench.py times); this class is its logic in Python, on top of
    MusicaProcessing, so that the choice of queues can be exercised without a GPU (tests/test_distributed.py).
    """

    HW_QUEUES = 4   # the HIP runtime's default number of hardware queues per process (GPU_MAX_HW_QUEUES)

    def __init__(self, image_size, levels=0, batch=1, depth=3, flags=0, device=0, calibrate=True):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.contexts = []
        if depth > 1:
            flags |= FLAG_LINEAR
        # Which hardware queue a stream lands on is the runtime's round-robin over every stream the process has created so far
        # (torch, RCCL, ...), and the queues are not equal: on MI355X two of the four share a pipe and do not run side by side
        # (three contexts on queues {0,1,2}: 0.364 ms per C4 step, on {1,2,3} or {2,3,0}: 0.407; two contexts on {2,3}: 0.455 — as
        # slow as one). So one context per queue is created, prime() times every cyclic window of `depth` of them for a few
        # steps and keeps the fastest; the others are destroyed.
        n = self.HW_QUEUES if (calibrate and 1 < depth < self.HW_QUEUES) else depth
        for _ in range(n):
            p = MusicaProcessing(device=device)
            if not p.init(image_size, levels=levels, batch=batch, flags=flags):
                self.cleanup()
                raise RuntimeError("musica_create failed: " + last_error())
            self.contexts.append(p)
        self.depth = depth
        self.batch = batch
        self.steps = 0
        self.calibration = None   # {first context of the window: ms per step} once prime() has chosen
        self._images = None

    def upload(self, images):
        """The shard resident in every context's input buffer (one array for all, or one per context in flight)."""
        per_ctx = isinstance(images, (list, tuple))
        if per_ctx and len(images) != self.depth:
            raise ValueError("need one image batch per context in flight")
        self._images = images
        for k, p in enumerate(self.contexts):
            p.upload(images[k % self.depth] if per_ctx else images)

    def _run(self, use, steps):
        for s in range(steps):
            if not use[s % len(use)].execute_device():
                raise RuntimeError("musica_execute_device failed: " + last_error())
        for p in use:
            p.sync()

    def prime(self, calibration_steps=9):
        """Two untimed steps per context (the first captures its graph, the second replays it), the choice of the contexts
        that stay (see __init__), then drain."""
        import time
        for _ in range(2):
            self._run(self.contexts, len(self.contexts))
        n = len(self.contexts)
        if n > self.depth:
            timing = {}
            for first in range(n):
                use = [self.contexts[(first + k) % n] for k in range(self.depth)]
                self._run(use, self.depth)                       # warm this combination
                best = None
                for _ in range(2):
                    t0 = time.perf_counter()
                    self._run(use, calibration_steps)
                    dt = (time.perf_counter() - t0) / calibration_steps * 1e3
                    best = dt if best is None else min(best, dt)
                timing[first] = round(best, 4)
            first = min(timing, key=timing.get)
            keep = [self.contexts[(first + k) % n] for k in range(self.depth)]
            for p in self.contexts:
                if not any(p is q for q in keep):
                    p.cleanup()
            self.contexts = keep
            self.calibration = timing
            if isinstance(self._images, (list, tuple)):          # one batch per context in flight: context k holds images[k]
                for k, p in enumerate(self.contexts):
                    p.upload(self._images[k])
                self._run(self.contexts, self.depth)
        self.steps = 0

    def step(self, d_pixels=None):
        """Enqueue one step (asynchronous) on the next context; returns that context."""
        p = self.contexts[self.steps % self.depth]
        if not p.execute_device(d_pixels):
            raise RuntimeError("musica_execute_device failed: " + last_error())
        self.steps += 1
        return p

    def last(self):
        """The context that ran the most recent step."""
        return self.contexts[(self.steps - 1) % self.depth]

    def sync(self):
        for p in self.contexts:
            p.sync()

    def cleanup(self):
        for p in self.contexts:
            p.cleanup()
        self.contexts = []
