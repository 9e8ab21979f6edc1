"""Multi-GPU sharding of independent utterances.

The hot path has no exchange step: every utterance carries its own predictor / vocoder state,
PCM history and RNG key, weights and codebooks are read-only replicas.  So the only
"parallelism" is a contiguous split of the utterance list over one process per GPU, and the
only collective is the end-of-run gather of a tiny report (timings, sample counts, codebook
usage histograms for the bitrate figure) - RCCL over xGMI on GPUs, gloo in the CPU rehearsal.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """contiguous block of utterances owned by `rank` (first ranks take the remainder)"""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_report(elapsed_s, samples, hist=None, device=None):
    """all-reduce the per-rank record: max elapsed, total samples, summed histograms.  Works with
    whichever torch.distributed backend is initialised (nccl == RCCL on ROCm, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return {"elapsed_s": float(elapsed_s), "samples": int(samples),
                "hist": None if hist is None else np.asarray(hist)}
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([samples], dtype=torch.int64, device=dev)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    out = {"elapsed_s": float(t.item()), "samples": int(s.item()), "hist": None}
    if hist is not None:
        h = torch.as_tensor(np.asarray(hist, dtype=np.int64)).to(dev)
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        out["hist"] = h.cpu().numpy()
    return out


def gather_records(record):
    """every rank's small dict (rank, device identity, shard, elapsed) on every rank, in rank order"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [record]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, record)
    return out
