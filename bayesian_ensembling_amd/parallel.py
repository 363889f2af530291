"""Member sharding over the GPUs of one node.

Ensemble members are independent until the weights are normalised over the model axis
(ensembles/data.py:391, weights.py:88,118-123), so each rank (one process per GPU) fits a contiguous block of
members and produces their rows of mean log-likelihood / posterior mean / posterior variance; ONE all-gather of
the packed [M_loc, 3, P] rows gives every rank the full [M, 3, P] table, after which normalisation and the
barycentre are computed redundantly on every rank.  On the GPU the gather is ncclAllGather (RCCL over xGMI) on
device rows through libbayens; `HostGather` runs the same exchange on host arrays over a torch.distributed
process group (gloo) and exists for the CPU test-suite and for moving the 128-byte RCCL unique id.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_members: int, rank: int, world: int):
    """Contiguous block partition; the first (n % world) ranks take one extra member."""
    base, extra = divmod(n_members, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def padded_rows(n_members: int, world: int) -> int:
    """Rows every rank contributes to the fixed-size all-gather (= ceil(M / world))."""
    return -(-n_members // world)


def pack_rows(ll_mean, mean, var, rows: int):
    """[M_loc, P] x3 -> [rows, 3, P] with NaN padding for ranks that own fewer members."""
    M_loc, P = ll_mean.shape
    out = np.full((rows, 3, P), np.nan)
    out[:M_loc, 0], out[:M_loc, 1], out[:M_loc, 2] = ll_mean, mean, var
    return out


def unpack_rows(gathered, n_members: int, world: int):
    """[world*rows, 3, P] -> (ll_mean, mean, var) each [M, P], dropping the padding rows."""
    rows = gathered.shape[0] // world
    keep = []
    for r in range(world):
        lo, hi = shard_bounds(n_members, r, world)
        keep.extend(range(r * rows, r * rows + (hi - lo)))
    g = gathered[keep]
    return g[:, 0], g[:, 1], g[:, 2]


class HostGather:
    """All-gather of equally sized host arrays over a torch.distributed group (gloo)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather(self, local: np.ndarray) -> np.ndarray:
        import torch
        t = torch.from_numpy(np.ascontiguousarray(local))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t, group=self.group)
        return np.concatenate([o.numpy() for o in outs], axis=0)

    def broadcast_bytes(self, payload: bytes, n: int, src: int = 0) -> bytes:
        import torch
        t = torch.zeros(n, dtype=torch.uint8)
        if self.rank == src:
            t[:] = torch.frombuffer(bytearray(payload[:n].ljust(n, b"\0")), dtype=torch.uint8)
        self.dist.broadcast(t, src=src, group=self.group)
        return bytes(t.numpy().tobytes())

    def max_float(self, x: float) -> float:
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t[0])

    def barrier(self):
        self.dist.barrier(group=self.group)


def init_rccl(ctx, host: HostGather):
    """Rank 0 draws the ncclUniqueId, everyone receives it over the host group, then joins the communicator."""
    uid = ctx.comm_unique_id() if host.rank == 0 else b""
    uid = host.broadcast_bytes(uid, 128, src=0)
    ctx.comm_init(host.rank, host.world, uid)


def weighted_posterior_sharded(n_members, local_ll, local_mean, local_var, gather, normalise, barycentre,
                               c_std=1.0):
    """The exchange step + the replicated tail, written once for both transports.

    gather(packed [rows,3,P]) -> [world*rows,3,P]; normalise(ll [M,P], c) -> w [M,P];
    barycentre(w, mean, var) -> (mu [P], var [P], n_bad).  Returns (w, mu, var)."""
    world = gather.world
    rows = padded_rows(n_members, world)
    table = gather.allgather(pack_rows(local_ll, local_mean, local_var, rows))
    ll, mean, var = unpack_rows(table, n_members, world)
    w = normalise(np.ascontiguousarray(ll), c_std)
    mu, v, _ = barycentre(w, np.ascontiguousarray(mean), np.ascontiguousarray(var))
    return w, mu, v


def w2_single_sharded(ctx, n_members, mu_local, chol_local):
    """ModelSimilarityWeight(mode='single') over member-sharded posteriors.

    mu_local [M_loc, N] / chol_local [M_loc, N, N]: this rank's members (equal M_loc on every rank, contiguous blocks in
    rank order).  The factors are all-gathered over RCCL (M_loc * N^2 doubles per rank), every rank runs the polar
    iterations of its share of the pairs and an all-reduce completes the distance table.  Returns (w [M], D [M, M])."""
    M_loc, N = mu_local.shape
    d_mu_loc, d_ch_loc = ctx.to_device(mu_local), ctx.to_device(chol_local)
    d_mu, d_ch = ctx.empty((n_members, N)), ctx.empty((n_members, N, N))
    d_w, d_D = ctx.empty((n_members,)), ctx.empty((n_members, n_members))
    try:
        ctx.allgather_rows(d_mu_loc, d_mu, M_loc, N)
        ctx.allgather_rows(d_ch_loc, d_ch, M_loc, N * N)
        ctx.w2_single_sharded_dev(n_members, N, d_mu, d_ch, d_w, d_D)
        return d_w.download(), d_D.download()
    finally:
        for d in (d_mu_loc, d_ch_loc, d_mu, d_ch, d_w, d_D):
            d.free()
