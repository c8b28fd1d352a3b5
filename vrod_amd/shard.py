"""Row sharding + the per-shard top-k exchange.

The corpus splits into contiguous row ranges, one per rank (one process per GPU).  Every
rank scans its range for the same query batch; the only exchange is one all-gather of
nq x k (id, score) pairs per rank -- 120 KB/rank at 1024 x 10, latency-bound on xGMI --
followed by an exact merge (ties -> smaller global id), so the result equals the 1-GPU
answer.  torch.distributed is plumbing: backend "nccl" is RCCL on ROCm; the CPU tests run
the same functions over "gloo".
"""
from __future__ import annotations


def shard_range(n_total: int, rank: int, world: int):
    """Rows [lo, hi) owned by `rank`; global id = local row + lo."""
    return n_total * rank // world, n_total * (rank + 1) // world


def all_gather_topk(dist, ids, scores, out_ids=None, out_scores=None):
    """ids/scores: [nq, k] tensors of this rank (int64 bits of u64 / float32), same shape on
    every rank.  Returns list-major [world, nq, k] tensors on every rank."""
    import torch
    world = dist.get_world_size()
    nq, k = ids.shape
    if out_ids is None:
        out_ids = torch.empty((world, nq, k), dtype=ids.dtype, device=ids.device)
    if out_scores is None:
        out_scores = torch.empty((world, nq, k), dtype=scores.dtype, device=scores.device)
    dist.all_gather_into_tensor(out_ids.view(-1), ids.contiguous().view(-1))
    dist.all_gather_into_tensor(out_scores.view(-1), scores.contiguous().view(-1))
    return out_ids, out_scores


def alloc_packed(nq: int, k: int, device):
    """One rank's result block: nq*k ids (u64, viewed as int64) immediately followed by nq*k
    scores (f32) in ONE byte buffer, so the exchange is a single all-gather.  Returns
    (packed uint8 [12*nq*k], ids view [nq,k] int64, scores view [nq,k] float32)."""
    import torch
    n = nq * k
    if n % 2:
        raise ValueError("nq*k must be even for the packed layout")
    packed = torch.empty(12 * n, dtype=torch.uint8, device=device)
    ids = packed[: 8 * n].view(torch.int64).view(nq, k)
    scores = packed[8 * n:].view(torch.float32).view(nq, k)
    return packed, ids, scores


def all_gather_packed(dist, packed, out=None):
    """packed: this rank's block (alloc_packed).  Returns [world * len(packed)] uint8: the
    blocks of all ranks in rank order (the input of vrod_merge_topk_packed_device)."""
    import torch
    world = dist.get_world_size()
    if out is None:
        out = torch.empty(world * packed.numel(), dtype=torch.uint8, device=packed.device)
    if packed.is_cuda and dist.get_backend() != "nccl":
        # rehearsal backends (gloo): stage through the host
        torch.cuda.current_stream(packed.device).synchronize()
        h_in = packed.cpu()
        h_out = torch.empty(world * packed.numel(), dtype=torch.uint8)
        dist.all_gather_into_tensor(h_out, h_in)
        out.copy_(h_out)
        return out
    dist.all_gather_into_tensor(out, packed)
    return out


def unpack_gathered(gathered, world: int, nq: int, k: int):
    """CPU-side view of a gathered packed buffer -> (ids uint64 [world,nq,k], scores f32 [world,nq,k])."""
    import numpy as np
    n = nq * k
    a = gathered.cpu().numpy().reshape(world, 12 * n)
    ids = np.ascontiguousarray(a[:, : 8 * n]).view(np.uint64).reshape(world, nq, k)
    sc = np.ascontiguousarray(a[:, 8 * n:]).view(np.float32).reshape(world, nq, k)
    return ids, sc
