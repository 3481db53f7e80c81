"""Batch sharding across the GPUs of one node (one process per GPU).

Both hot paths are embarrassingly parallel over utterances (MLPG: one banded
system per (utterance, static dim); DTW: one alignment per utterance pair), so
the multi-GPU form is: every rank processes a contiguous slice of the batch with
the single-GPU kernels and NO collective on the data path; the only exchange is
an optional gather of the results (``torch.distributed`` all-gather: RCCL over
xGMI with the ``nccl`` backend on GPUs, ``gloo`` on CPU in the tests).

The reference has no counterpart: it is a single-process library and batches by
a Python loop (/root/reference/nnmnkwii/util/__init__.py:44-66).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, balanced slice [lo, hi) of ``n`` items for ``rank`` of ``world``
    (the first ``n % world`` ranks get one extra item)."""
    assert world >= 1 and 0 <= rank < world
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    hi = lo + base + (1 if rank < extra else 0)
    return lo, hi


def _dist():
    import torch.distributed as dist
    return dist


def _world(group=None):
    dist = _dist()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def all_gather_shards(local, n_total, group=None):
    """Gather per-rank shards (leading axis) into the full array on every rank.

    ``local`` is a torch tensor (on the GPU for ``nccl``, on the CPU for ``gloo``)
    holding this rank's ``shard_range`` rows.  Shards may differ by one row, so
    they are padded to the largest shard for a single ``all_gather_into_tensor``.
    """
    import torch
    dist = _dist()
    rank, world = _world(group)
    if world == 1:
        return local
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    parts = [out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(sizes)]
    return torch.cat(parts, dim=0)


def _gather_sizes(n, device, group=None):
    """[n of rank 0, n of rank 1, ...] on every rank."""
    import torch
    dist = _dist()
    rank, world = _world(group)
    if world == 1:
        return [int(n)]
    mine = torch.tensor([n], dtype=torch.int64, device=device)
    ns = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(ns, mine, group=group)
    return [int(x) for x in ns.cpu().tolist()]


def all_gather_varsize(local, group=None):
    """Gather shards whose leading sizes are known only to their owners (each rank loaded its own part of the
    corpus): one small all-gather of the sizes, then one padded ``all_gather_into_tensor`` of the data.
    Returns (full tensor in rank order, list of per-rank sizes)."""
    import torch
    dist = _dist()
    rank, world = _world(group)
    if world == 1:
        return local, [int(local.shape[0])]
    sizes = _gather_sizes(int(local.shape[0]), local.device, group)
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad.contiguous(), group=group)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0), sizes


def mlpg_batch_sharded(means, variances, windows, lengths=None, gather=True, group=None, compute=None,
                       local_shards=False):
    """MLPG over a ``(B, Tmax, D)`` batch split across the ranks of ``group``.

    ``local_shards=False``: every rank passes the same full arrays (numpy or torch); rank ``r`` runs
    ``paramgen.mlpg_batch`` on utterances ``shard_range(B, r, world)`` on its own
    GPU.  With ``gather=True`` the ``(B, Tmax, sd)`` result is all-gathered and
    returned on every rank (numpy in -> numpy out); with ``gather=False`` only
    the local shard is returned together with its ``(lo, hi)`` range.

    ``local_shards=True``: every rank passes ONLY its own utterances (the way a multi-process data loader hands
    them out: no rank ever holds the whole batch); shard sizes may differ (even be zero).  ``gather=True``
    concatenates the results in rank order on every rank (ranks may have padded their shards to different ``Tmax``:
    the results are grown to the global maximum before the gather; the static dimension must agree);
    ``gather=False`` returns the local result and this rank's ``(lo, hi)`` position in that order.

    ``compute(means, variances, windows, lengths)`` overrides the per-shard
    kernel call (the CPU tests inject a checker there; the default is the HIP path).
    """
    import torch
    rank, world = _world(group)
    B = means.shape[0]
    lo, hi = (0, B) if local_shards else shard_range(B, rank, world)
    if compute is None:
        from .paramgen import mlpg_batch as compute
    is_np = not torch.is_tensor(means)
    m = means[lo:hi]
    v = variances
    if v is not None and v.ndim == 3:
        v = v[lo:hi]
    L = None if lengths is None else lengths[lo:hi]
    sd = means.shape[2] // len(windows)
    if hi > lo:
        y = compute(m, v, windows, L)
    else:
        y = (np.zeros((0, means.shape[1], sd), dtype=means.dtype) if is_np
             else means.new_zeros((0, means.shape[1], sd)))
    dist = _dist()
    if not gather and not local_shards:
        return y, (lo, hi)
    yt = torch.from_numpy(np.ascontiguousarray(y)) if is_np else y
    if world > 1 and dist.get_backend(group) == "nccl" and not yt.is_cuda:
        yt = yt.cuda()
    if local_shards:
        if not gather:
            sizes = _gather_sizes(int(yt.shape[0]), yt.device, group)
            start = sum(sizes[:rank])
            return y, (start, start + sizes[rank])
        # ranks that padded their own shards may disagree on Tmax: pad every shard to the global maximum first
        # (all_gather_into_tensor needs identical trailing shapes; the padding frames of an MLPG result are zero anyway)
        if world > 1:
            tmax = torch.tensor([yt.shape[1], yt.shape[2], -yt.shape[2]], dtype=torch.int64, device=yt.device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=group)
            t_glob, sd_max, sd_min = int(tmax[0]), int(tmax[1]), -int(tmax[2])
            if sd_max != sd_min:
                raise ValueError("mlpg_batch_sharded(local_shards=True): ranks disagree on the static dimension "
                                 "(%d vs %d)" % (sd_min, sd_max))
            if t_glob != yt.shape[1]:
                grown = yt.new_zeros((yt.shape[0], t_glob, yt.shape[2]))
                grown[:, : yt.shape[1]] = yt
                yt = grown
        full = all_gather_varsize(yt, group)[0]
    else:
        full = all_gather_shards(yt, B, group)
    return full.cpu().numpy() if is_np else full


def dtw_align_sharded(aligner, X, Y, group=None, transform=None, local_shards=False):
    """``DTWAligner.transform`` over pairs split across the ranks of ``group``.

    Every rank aligns pairs ``shard_range(N, r, world)`` of the full arrays -- or, with ``local_shards=True``, the
    pairs it was handed (no rank holds the whole corpus; shard sizes may differ); the outputs are padded
    to the global maximum length (the reference grows its outputs to the longest
    warping path of the WHOLE batch, alignment.py:55-71) and all-gathered in rank order.
    numpy in -> numpy out on every rank.
    """
    import torch
    dist = _dist()
    rank, world = _world(group)
    N = X.shape[0]
    lo, hi = (0, N) if local_shards else shard_range(N, rank, world)
    if transform is None:
        transform = aligner.transform
    longer = X if X.shape[1] > Y.shape[1] else Y
    D = longer.shape[2]
    if hi > lo:
        Xa, Ya = transform((X[lo:hi], Y[lo:hi]))
    else:
        Xa = np.zeros((0, longer.shape[1], D), dtype=longer.dtype)
        Ya = np.zeros_like(Xa)
    if world == 1:
        return Xa, Ya
    use_cuda = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    tlen = torch.tensor([Xa.shape[1]], dtype=torch.int64, device=dev)
    dist.all_reduce(tlen, op=dist.ReduceOp.MAX, group=group)
    T_out = int(tlen.item())

    def pad(a):
        # upload the shard as it is and grow it on the device: no host-side padded copy
        t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        if t.shape[1] == T_out:
            return t
        out = torch.zeros((a.shape[0], T_out, D), dtype=t.dtype, device=dev)
        out[:, : a.shape[1]] = t
        return out

    if local_shards:
        return all_gather_varsize(pad(Xa), group)[0].cpu().numpy(), all_gather_varsize(pad(Ya), group)[0].cpu().numpy()
    Xf = all_gather_shards(pad(Xa), N, group).cpu().numpy()
    Yf = all_gather_shards(pad(Ya), N, group).cpu().numpy()
    return Xf, Yf
