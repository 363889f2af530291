"""Member sharding over the GPUs of one node.

Ensemble members are independent until the weights are normalised over the model axis
(ensembles/data.py:391, weights.py:88,118-123), so each rank (one process per GPU) fits a contiguous block of
members and produces their rows of mean log-likelihood / posterior mean / posterior variance; ONE all-gather of
the packed [rows, 3, P] rows gives every rank the full [M, P] tables, after which normalisation and the
barycentre are computed redundantly on every rank.  M need not divide by the number of ranks.

`weighted_posterior_sharded` is that exchange + tail, written once over an *exchange* object:

  RcclExchange   device rows, ncclAllGather (RCCL over xGMI) inside libbayens - the product path, what bench.py's
                 N > 1 leg runs.  No PyTorch: the 128-byte RCCL id travels through a FileRendezvous, barriers and the
                 max-over-ranks of the timing protocol are an ncclAllReduce (bayens_comm_max).
  HostGather     host arrays over a torch.distributed process group (gloo) - the CPU test-suite's transport.
  FileRendezvous host arrays through a shared directory - the bootstrap of RcclExchange, and a rehearsal transport
                 for several ranks on ONE GPU (RCCL refuses two ranks on one device).
"""
from __future__ import annotations

import os
import pickle
import time

import numpy as np


def shard_bounds(n_members: int, rank: int, world: int):
    """Contiguous block partition; the first (n % world) ranks take one extra member (= bayens_shard_bounds)."""
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
    return (np.ascontiguousarray(g[:, 0]), np.ascontiguousarray(g[:, 1]), np.ascontiguousarray(g[:, 2]))


class _HostRows:
    """gather_rows for transports that all-gather equally sized host arrays (`allgather`)."""

    def gather_rows(self, n_members, ll_loc, mean_loc, var_loc):
        rows = padded_rows(n_members, self.world)
        table = self.allgather(pack_rows(ll_loc, mean_loc, var_loc, rows))
        return unpack_rows(table, n_members, self.world)


class HostGather(_HostRows):
    """Host arrays over a torch.distributed group (gloo): the CPU test-suite's transport."""

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


class FileRendezvous(_HostRows):
    """All-gather of small host payloads through a directory every rank can see (one node): each round, rank r
    writes `<dir>/<round>.<r>` atomically (temp file + rename) and polls for the others'.  Carries the 128-byte RCCL
    id before the communicator exists; with `gather_rows` it is also a complete (slow) exchange for rehearsing several
    ranks on one GPU.  No third-party dependency."""

    def __init__(self, directory: str, rank: int, world: int, timeout_s: float = 300.0):
        self.dir, self.rank, self.world, self.timeout_s = directory, int(rank), int(world), timeout_s
        self.round = 0
        os.makedirs(directory, exist_ok=True)

    @staticmethod
    def directory_from_env():
        """BAYENS_RDZV (set by bench.py when it spawns the ranks itself), else a directory keyed by the launcher:
        all ranks of one `torch.distributed.run` share MASTER_PORT and their parent's pid."""
        d = os.environ.get("BAYENS_RDZV")
        if d:
            return d
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bayens_rdzv_{key}")

    def allgather_obj(self, obj):
        tag = f"{self.round:06d}"
        self.round += 1
        mine = os.path.join(self.dir, f"{tag}.{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)
        os.replace(mine + ".tmp", mine)
        out, deadline = [], time.time() + self.timeout_s
        for r in range(self.world):
            path = os.path.join(self.dir, f"{tag}.{r}")
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never reached rendezvous round {tag} in {self.dir}")
                time.sleep(0.002)
            with open(path, "rb") as f:
                out.append(pickle.load(f))
        return out

    def allgather(self, local: np.ndarray) -> np.ndarray:
        return np.concatenate(self.allgather_obj(np.ascontiguousarray(local)), axis=0)

    def broadcast_bytes(self, payload: bytes, n: int, src: int = 0) -> bytes:
        return bytes(self.allgather_obj(bytes(payload[:n]) if self.rank == src else b"")[src]).ljust(n, b"\0")

    def max_float(self, x: float) -> float:
        return float(max(self.allgather_obj(float(x))))

    def barrier(self):
        self.allgather_obj(None)

    def close(self):
        """Final barrier; every other rank then leaves a `done` marker once it has read that round, and rank 0
        removes the directory when all markers are there (so no rank can find a file gone under its feet)."""
        self.barrier()
        if self.rank != 0:
            open(os.path.join(self.dir, f"done.{self.rank}"), "wb").close()
            return
        deadline = time.time() + self.timeout_s
        while any(not os.path.exists(os.path.join(self.dir, f"done.{r}")) for r in range(1, self.world)):
            if time.time() > deadline:
                return
            time.sleep(0.002)
        for name in os.listdir(self.dir):
            try:
                os.remove(os.path.join(self.dir, name))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)
        except OSError:
            pass


class RcclExchange:
    """Device rows over RCCL (ncclAllGather / ncclAllReduce inside libbayens, over xGMI).  `bootstrap` is any object
    with rank / world / broadcast_bytes (FileRendezvous, HostGather): rank 0 draws the ncclUniqueId, everyone
    receives it, then joins the communicator."""

    def __init__(self, ctx, bootstrap):
        self.ctx, self.rank, self.world = ctx, bootstrap.rank, bootstrap.world
        uid = ctx.comm_unique_id() if self.rank == 0 else b""
        uid = bootstrap.broadcast_bytes(uid, 128, src=0)
        ctx.comm_init(self.rank, self.world, uid)
        self._out = {}

    def gather_rows(self, n_members, ll_loc, mean_loc, var_loc):
        """Device buffers [M_loc, P] x3 -> device buffers [M, P] x3 (owned by the exchange, reused per shape)."""
        lo, hi = shard_bounds(n_members, self.rank, self.world)
        P = ll_loc.shape[-1]
        out = self._out.setdefault((n_members, P), tuple(self.ctx.empty((n_members, P)) for _ in range(3)))
        self.ctx.allgather_member_rows(n_members, P, hi - lo, ll_loc, mean_loc, var_loc, *out)
        return out

    def max_float(self, x: float) -> float:
        return self.ctx.comm_max(x)

    def barrier(self):
        self.ctx.comm_max(0.0)

    def close(self):
        self.ctx.comm_destroy()


def make_exchange(ctx, rdzv, prefer="rccl"):
    """The rows exchange of a multi-rank run: RCCL when every rank can load it and the communicator comes up, else the
    host-staged transport over the rendezvous - decided by ALL ranks together, so that no rank waits inside
    ncclCommInitRank for a peer that has already given up.  Returns (exchange, kind, note)."""
    if prefer != "rccl":
        return HostStagedExchange(ctx, rdzv), "file", None
    try:                                   # phase 1: can this rank load librccl at all?  (no communicator yet)
        ctx.comm_unique_id()
        mine = (True, "")
    except Exception as e:                 # noqa: BLE001 - whatever it is, the peers must hear about it
        mine = (False, f"rank {rdzv.rank}: {e}")
    flags = rdzv.allgather_obj(mine)
    if all(f[0] for f in flags):
        try:                               # phase 2: bring the communicator up (a failure here is symmetric)
            ex = RcclExchange(ctx, rdzv)
            mine = (True, "")
        except Exception as e:             # noqa: BLE001
            ex, mine = None, (False, f"rank {rdzv.rank}: {e}")
        flags = rdzv.allgather_obj(mine)
        if all(f[0] for f in flags):
            return ex, "rccl", None
        if ex is not None:
            ex.close()
    note = next(f[1] for f in flags if not f[0])
    return HostStagedExchange(ctx, rdzv), "file", "RCCL unavailable, host-staged exchange used instead: " + note


class HostStagedExchange:
    """RcclExchange's interface (device rows in, device tables out) carried over a host transport: rows are
    downloaded, gathered by `host` (FileRendezvous / HostGather) and uploaded.  A rehearsal transport - several ranks
    on ONE GPU, where RCCL refuses to build a communicator - never the measured path."""

    def __init__(self, ctx, host):
        self.ctx, self.host, self.rank, self.world = ctx, host, host.rank, host.world
        self._out = {}

    def gather_rows(self, n_members, ll_loc, mean_loc, var_loc):
        P = ll_loc.shape[-1]
        lo, hi = shard_bounds(n_members, self.rank, self.world)
        rows = [b.download((hi - lo, P)) for b in (ll_loc, mean_loc, var_loc)]
        tables = self.host.gather_rows(n_members, *rows)
        out = self._out.setdefault((n_members, P), tuple(self.ctx.empty((n_members, P)) for _ in range(3)))
        for d, h in zip(out, tables):
            d.upload(h)
        return out

    def max_float(self, x):
        return self.host.max_float(x)

    def barrier(self):
        self.ctx.sync()
        self.host.barrier()

    def close(self):
        pass


def init_rccl(ctx, bootstrap):
    """Kept for callers of the round-1 name: joins the communicator, returns the RcclExchange."""
    return RcclExchange(ctx, bootstrap)


def weighted_posterior_sharded(n_members, local_ll, local_mean, local_var, exchange, normalise, barycentre,
                               c_std=1.0, weights=None):
    """The exchange step + the replicated tail of the weighted posterior, written once for every transport
    (ensembles/weights.py:118-123 normalisation over ALL members, then ensemble_scheme.py:52-78).

    exchange.gather_rows(M, ll, mean, var) -> the three [M, P] tables (host arrays or device buffers, as the
    transport carries them); normalise(ll [M,P], c) -> w [M,P]; barycentre(w, mean, var) -> (mu [P], var [P], n_bad).
    weights(ll, mean, var) -> w, when given, replaces normalise: weights that are functions of the members' moments
    (ModelSimilarityWeight mode='temporal', weights.py:302-325) need the same three tables and nothing more.
    EVERY rank must call this (the gather is a collective).  Returns (w, mu, var)."""
    ll, mean, var = exchange.gather_rows(n_members, local_ll, local_mean, local_var)
    w = weights(ll, mean, var) if weights is not None else normalise(ll, c_std)
    mu, v, _ = barycentre(w, mean, var)
    return w, mu, v


def w2_single_sharded(ctx, n_members, mu_local, chol_local):
    """ModelSimilarityWeight(mode='single') over member-sharded posteriors.

    mu_local [M_loc, N] / chol_local [M_loc, N, N]: this rank's members - the SAME M_loc on every rank (contiguous
    blocks in rank order), i.e. n_members == nranks * M_loc; pad the ensemble or use the single-GPU entry otherwise.
    The factors are all-gathered over RCCL (M_loc * N^2 doubles per rank), every rank runs the polar iterations of its
    share of the pairs and an all-reduce completes the distance table.  Returns (w [M], D [M, M])."""
    M_loc, N = mu_local.shape
    _, world = ctx.comm_info()[:2]
    if world <= 0:
        raise ValueError("w2_single_sharded: the context has no communicator (RcclExchange / comm_init first)")
    if n_members != world * M_loc:
        raise ValueError(f"w2_single_sharded: n_members={n_members} must equal nranks*M_loc={world}*{M_loc} "
                         "(the fixed-size all-gather of the factors cannot carry uneven shards)")
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
