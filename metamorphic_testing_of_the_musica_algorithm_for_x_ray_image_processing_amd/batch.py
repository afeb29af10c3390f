"""Batch driver: independent images sharded one-shard-per-GPU, statistics gathered with one collective.

The reference processes one image per process on one device. A batch of independent images shards
trivially: rank r owns the images `assign_images(total, world)[r]`, runs them through its
own `MusicaProcessing` context with no data-path communication, and at the end one all-gather (RCCL over
xGMI on GPUs, gloo in the CPU tests) collects the fixed-size `musica_stats` row of every image.
Steps in flight on one GPU are the C ABI's musica_pipeline_* (processing.MusicaPipeline).
"""
import ctypes as C

import numpy as np

from .processing import Stats

STATS_WORDS = C.sizeof(Stats) // 4  # 17 x 32-bit words per image


def assign_images(total, world):
    """Image k goes to rank k mod world; returns the list of image ids of every rank."""
    return [list(range(r, total, world)) for r in range(world)]


def stats_to_row(st):
    """One `musica_stats` (ctypes) -> int32[STATS_WORDS] with the struct's exact bytes."""
    return np.frombuffer(bytes(st), dtype=np.int32).copy()


def row_to_stats(row):
    return Stats.from_buffer_copy(np.ascontiguousarray(row, dtype=np.int32).tobytes())


def gather_rows(local_rows, world, dist=None, device=None, force_collective=False):
    """All-gather equal-sized int32 row blocks; returns int32[world * rows, STATS_WORDS] on every rank.

    `local_rows` is a numpy int32 array (CPU / gloo) or a torch tensor already on `device` (GPU / RCCL).
    `force_collective`: run the collective even in a job of one rank (the one-GPU test of the RCCL path).
    """
    import torch
    if isinstance(local_rows, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(local_rows, dtype=np.int32))
        if device is not None:
            t = t.to(device)
    else:
        t = local_rows
    if dist is None or (world == 1 and not force_collective):
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
