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

import io
import json
import os
import re
import stat
import sys
import threading
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


def _encode(obj) -> bytes:
    """Fixed wire formats (no pickle: a rendezvous file is read by every rank, so it must not be able to carry code):
    b'B' + raw bytes | b'N' + .npy image (allow_pickle=False) | b'J' + JSON (None, bool, int, float, str, lists)."""
    if isinstance(obj, (bytes, bytearray)):
        return b"B" + bytes(obj)
    if isinstance(obj, np.ndarray):
        buf = io.BytesIO()
        np.save(buf, np.ascontiguousarray(obj), allow_pickle=False)
        return b"N" + buf.getvalue()
    return b"J" + json.dumps(obj).encode()


def _decode(raw: bytes):
    kind, body = raw[:1], raw[1:]
    if kind == b"B":
        return body
    if kind == b"N":
        return np.load(io.BytesIO(body), allow_pickle=False)
    if kind == b"J":
        return json.loads(body.decode())
    raise ValueError("rendezvous file with an unknown format tag")


class FileRendezvous(_HostRows):
    """All-gather of small host payloads through a directory every rank can see (one node): each round, rank r
    writes `<dir>/<session>.<round>.<r>` atomically (temp file + rename) and polls for the others'.  Carries the
    128-byte RCCL id before the communicator exists; with `gather_rows` it is also a complete (slow) exchange for
    rehearsing several ranks on one GPU.  No third-party dependency.

    Safety of the directory (it may sit under a shared /tmp): it is created with mode 0700, and one that already
    exists is accepted only if this user owns it and nobody else can write to it.  Payloads travel in fixed formats
    (raw bytes / .npy without pickle / JSON), never pickle.  Every attempt has its own SESSION token - rank 0 clears the
    directory, draws the token and publishes it in `session.<restart count>` together with the attempt's NONCE (what the
    launcher gave every rank of this attempt and of no other: `attempt_nonce`); the other ranks take a session file only
    if it carries their own nonce - so neither the round files nor the session of an earlier attempt in the same
    directory (a worker restart under the same launcher, a recycled pid) are ever read."""

    def __init__(self, directory: str, rank: int, world: int, timeout_s: float = 300.0):
        self.dir, self.rank, self.world, self.timeout_s = directory, int(rank), int(world), timeout_s
        self.round = 0
        try:
            os.makedirs(directory, mode=0o700)
        except FileExistsError:
            pass
        st = os.lstat(directory)                         # lstat: a symlink planted under a shared /tmp is not a directory
        if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode) or st.st_uid != os.geteuid() or (st.st_mode & 0o022):
            raise PermissionError(f"rendezvous directory {directory} is not a directory (no symlink) owned by this user that "
                                  "only this user can write to: refusing to exchange through it")
        self.session = self._open_session()

    @staticmethod
    def attempt_nonce() -> str:
        """What every rank of ONE launch attempt shares and no other attempt does: BAYENS_RDZV_NONCE when the launcher set one
        (bench.py's spawn_ranks draws a random one per run), else the launcher's run id, restart count, port and pid."""
        n = os.environ.get("BAYENS_RDZV_NONCE")
        if n:
            return n
        # (the parent's pid is part of the fallback only under torch's launcher, whose ranks share their parent; under srun / mpirun
        #  wrappers they need not - set BAYENS_RDZV_NONCE there, or every non-zero rank waits for a session that never matches)
        shared_parent = "TORCHELASTIC_RUN_ID" in os.environ
        return ":".join([os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"),
                         os.environ.get("MASTER_PORT", "0"), str(os.getppid()) if shared_parent else "-"])

    def _open_session(self) -> str:
        """The session token of THIS attempt.  Rank 0 clears what earlier attempts left in the directory, draws a token and
        publishes it with the attempt's nonce; the others accept a session file only if it carries the SAME nonce (round 4:
        the file used to be tied to rank 0's pid being alive, which a recycled pid, a zombie or a still-running earlier
        rank 0 all satisfy - a peer then adopted the old token, read the old attempt's ncclUniqueId and sat in
        ncclCommInitRank until the watchdog fired)."""
        restart = os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
        path = os.path.join(self.dir, f"session.{restart}")
        nonce = self.attempt_nonce()
        if self.rank == 0:
            self._remove_own_files()                     # stale session / round / done files of earlier attempts
            token = f"{restart}-{os.urandom(8).hex()}"
            with open(path + ".tmp", "wb") as f:
                f.write(_encode({"token": token, "pid": os.getpid(), "nonce": nonce}))
            os.replace(path + ".tmp", path)
            return token
        deadline = time.time() + self.timeout_s
        while True:
            try:
                with open(path, "rb") as f:
                    info = _decode(f.read())
                if str(info["nonce"]) == nonce:          # this attempt's rank 0 wrote it
                    return str(info["token"])
            except (OSError, ValueError, KeyError, TypeError):
                pass                                     # no file yet, a torn read, or another attempt's file: wait
            if time.time() > deadline:
                raise TimeoutError(f"rank {self.rank}: rank 0 never opened a rendezvous session of this attempt in {self.dir}")
            time.sleep(0.002)

    @staticmethod
    def directory_from_env():
        """BAYENS_RDZV (set by bench.py when it spawns the ranks itself), else a directory keyed by the launcher:
        all ranks of one `torch.distributed.run` share MASTER_PORT and their parent's pid."""
        d = os.environ.get("BAYENS_RDZV")
        if d:
            return d
        key = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}"
        return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"bayens_rdzv_{os.geteuid()}_{key}")

    def allgather_obj(self, obj, timeout_s=None):
        tag = f"{self.session}.{self.round:06d}"
        self.round += 1
        mine = os.path.join(self.dir, f"{tag}.{self.rank}")
        with open(mine + ".tmp", "wb") as f:
            f.write(_encode(obj))
        os.replace(mine + ".tmp", mine)
        out, deadline = [], time.time() + (self.timeout_s if timeout_s is None else timeout_s)
        for r in range(self.world):
            path = os.path.join(self.dir, f"{tag}.{r}")
            while not os.path.exists(path):
                if time.time() > deadline:
                    raise TimeoutError(f"rank {self.rank}: rank {r} never reached rendezvous round {tag} in {self.dir}")
                time.sleep(0.002)
            with open(path, "rb") as f:
                out.append(_decode(f.read()))
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
            open(os.path.join(self.dir, f"{self.session}.done.{self.rank}"), "wb").close()
            return
        deadline = time.time() + self.timeout_s
        while any(not os.path.exists(os.path.join(self.dir, f"{self.session}.done.{r}")) for r in range(1, self.world)):
            if time.time() > deadline:
                return
            time.sleep(0.002)
        self._remove_own_files()
        try:
            os.rmdir(self.dir)                           # (only an empty directory goes: somebody else's files keep it)
        except OSError:
            pass

    # the names this class writes: session.<restart>, <restart>-<16 hex>.<round>.<rank>, <restart>-<16 hex>.done.<rank>, and their
    # .tmp forms.  Nothing else in the directory is ever touched: BAYENS_RDZV may name a directory that holds a user's files.
    _OWN = re.compile(r"^(session\.\d+|\d+-[0-9a-f]{16}\.(\d{6}|done)\.\d+)(\.tmp)?$")

    def _remove_own_files(self):
        for name in os.listdir(self.dir):
            if self._OWN.match(name):
                try:
                    os.remove(os.path.join(self.dir, name))
                except OSError:
                    pass


class CommInitWatchdog:
    """Bounds the time a rank may spend inside ncclCommInitRank.  RCCL's bootstrap is a blocking rendezvous of ALL
    ranks: if one peer failed before entering it (or died inside it), the others wait forever.  While the guarded block
    runs, a timer thread is armed; if it fires, the process reports and leaves with exit code 75 (`os._exit`: the main
    thread is stuck in a C call and cannot be interrupted), and the parent - bench.py's spawn_ranks or the launcher -
    terminates the remaining ranks.  A run that loses its communicator therefore ENDS, non-zero, instead of hanging."""

    EXIT_CODE = 75

    def __init__(self, rank, timeout_s, what="ncclCommInitRank", _exit=os._exit):
        self.rank, self.timeout_s, self.what, self._exit = rank, float(timeout_s), what, _exit
        self._timer = None

    def _fire(self):
        print(f"rank {self.rank}: {self.what} did not return within {self.timeout_s:.0f} s - a peer never joined the "
              "communicator; leaving with exit code %d" % self.EXIT_CODE, file=sys.stderr, flush=True)
        self._exit(self.EXIT_CODE)

    def __enter__(self):
        self._timer = threading.Timer(self.timeout_s, self._fire)
        self._timer.daemon = True
        self._timer.start()
        return self

    def __exit__(self, *exc):
        self._timer.cancel()
        return False


class RcclExchange:
    """Device rows over RCCL (ncclAllGather / ncclAllReduce inside libbayens, over xGMI).  `bootstrap` is any object
    with rank / world / broadcast_bytes (FileRendezvous, HostGather): rank 0 draws the ncclUniqueId, everyone
    receives it, then joins the communicator (bounded by CommInitWatchdog)."""

    def __init__(self, ctx, bootstrap, init_timeout_s=180.0):
        self.ctx, self.rank, self.world = ctx, bootstrap.rank, bootstrap.world
        uid = ctx.comm_unique_id() if self.rank == 0 else b""
        uid = bootstrap.broadcast_bytes(uid, 128, src=0)
        with CommInitWatchdog(self.rank, init_timeout_s):
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


def _allgather_with_timeout(rdzv, obj, timeout_s):
    """rdzv.allgather_obj with a bound of its own where the rendezvous supports one (FileRendezvous); HostGather (gloo) has the
    process group's timeout."""
    try:
        return rdzv.allgather_obj(obj, timeout_s=timeout_s)
    except TypeError:
        return rdzv.allgather_obj(obj)


class ExchangeUnavailable(RuntimeError):
    """RCCL could not be loaded / the communicator did not come up and the host-staged exchange was not allowed."""


def make_exchange(ctx, rdzv, prefer="rccl", allow_host_exchange=False, init_timeout_s=180.0):
    """The rows exchange of a multi-rank run, decided by ALL ranks together.  Returns (exchange, kind, note).

    prefer="rccl" (the product): RCCL, or an error.  Phase 1 is a LOAD-ONLY probe (`ctx.comm_available()`: dlopen + symbol
    lookup; no ncclUniqueId is drawn, so no bootstrap listener is left behind) whose result the ranks compare through the
    rendezvous: a rank that cannot load librccl stops everybody BEFORE anyone enters the blocking ncclCommInitRank.
    Phase 2 brings the communicator up under CommInitWatchdog and the ranks compare notes again.  A failure is fatal -
    `ExchangeUnavailable` on every rank that can still speak, exit code 75 from a rank stuck inside RCCL - unless
    `allow_host_exchange` (bench.py --allow-host-exchange) explicitly permits the host-staged transport: that one moves
    the rows through the rendezvous directory, is a rehearsal transport, and says so in config.exchange = "file".
    Phase 2 assumes nothing about symmetry: ncclCommInitRank failing on ONE rank leaves its peers blocked inside it,
    which is exactly what the watchdog (and the parent's terminate-on-first-failure) is for.

    prefer="file": the host-staged transport, asked for by name (several ranks rehearsing on ONE GPU)."""
    if prefer != "rccl":
        return HostStagedExchange(ctx, rdzv), "file", None
    try:                                   # phase 1: can this rank load librccl at all?  (no communicator, no id)
        ctx.comm_available()
        mine = [True, ""]
    except Exception as e:                 # noqa: BLE001 - whatever it is, the peers must hear about it
        mine = [False, f"rank {rdzv.rank}: {e}"]
    flags = rdzv.allgather_obj(mine)
    if all(f[0] for f in flags):
        try:                               # phase 2: bring the communicator up (bounded by the watchdog)
            ex = RcclExchange(ctx, rdzv, init_timeout_s)
            mine = [True, ""]
        except Exception as e:             # noqa: BLE001
            ex, mine = None, [False, f"rank {rdzv.rank}: {e}"]
        # A peer stuck inside ncclCommInitRank never reaches this round; its watchdog ends it after init_timeout_s.  A rank
        # that is waiting here for such a peer therefore gives up shortly after that - not after the rendezvous' own, longer
        # timeout - and leaves the documented way (ExchangeUnavailable -> exit code 3), with its own failure if it had one.
        try:
            flags = _allgather_with_timeout(rdzv, mine, init_timeout_s + 30.0)
        except TimeoutError as e:
            if ex is not None:
                ex.close()
            raise ExchangeUnavailable("RCCL exchange unavailable (" + (mine[1] or f"rank {rdzv.rank}: a peer never reported "
                                      "back from ncclCommInitRank") + f"; {e})") from e
        if all(f[0] for f in flags):
            return ex, "rccl", None
        if ex is not None:
            ex.close()
    note = next(f[1] for f in flags if not f[0])
    if not allow_host_exchange:
        raise ExchangeUnavailable("RCCL exchange unavailable (" + note + "); pass --allow-host-exchange / "
                                  "allow_host_exchange=True to rehearse over the host-staged transport instead")
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


def gather_member_blocks(ctx, n_members, rank, world, local, row_shape):
    """All-gather of per-rank blocks of leading-axis rows that need NOT have the same length: rank r contributes the rows
    shard_bounds(n_members, r, world) of a [n_members, *row_shape] table.  The collective itself is fixed-size
    (bayens_allgather_rows = one ncclAllGather): every rank sends ceil(n_members / world) rows - its own, then padding - and
    the gathered image [world][rows][...] is compacted on the device, block by block, into the [n_members][...] table
    (device-to-device copies; with even shards the image IS the table and nothing is copied).  Returns the device table;
    the caller frees it."""
    row_shape = tuple(int(x) for x in row_shape)
    width = int(np.prod(row_shape, dtype=np.int64)) if row_shape else 1
    rows = -(-n_members // world)
    lo, hi = shard_bounds(n_members, rank, world)
    local = np.ascontiguousarray(local, dtype=np.float64).reshape((hi - lo,) + row_shape)
    d_loc = ctx.empty((rows,) + row_shape)
    d_img = ctx.empty((world * rows,) + row_shape)
    try:
        d_loc.upload(local)                                  # (rows beyond this rank's share stay unwritten: never copied below)
        ctx.allgather_rows(d_loc, d_img, rows, width)
        if n_members == world * rows:
            out, d_img = d_img, None
            return out
        out = ctx.empty((n_members,) + row_shape)
        for r in range(world):
            lo_r, hi_r = shard_bounds(n_members, r, world)
            if hi_r > lo_r:
                out.copy_rows_from(lo_r, d_img, r * rows, hi_r - lo_r)
        return out
    finally:
        d_loc.free()
        if d_img is not None:
            d_img.free()


def w2_single_sharded(ctx, n_members, mu_local, chol_local):
    """ModelSimilarityWeight(mode='single') over member-sharded posteriors (weights.py:240-265 has no restriction
    on the member count).

    mu_local [M_loc, N] / chol_local [M_loc, N, N]: this rank's members - the block shard_bounds(n_members, rank, world) of the
    ensemble, so the shards may be uneven (the first n_members % world ranks hold one member more), exactly as the member
    exchange of the weighted posterior deals them.  The factors are all-gathered over RCCL (ceil(M / world) * N^2 doubles per
    rank, padded blocks compacted on the device), every rank runs the polar iterations of its share of the pairs and an
    all-reduce completes the distance table.  Returns (w [M], D [M, M])."""
    M_loc, N = mu_local.shape
    rank, world = ctx.comm_info()[:2]
    if world <= 0:
        raise ValueError("w2_single_sharded: the context has no communicator (RcclExchange / comm_init first)")
    lo, hi = shard_bounds(n_members, rank, world)
    if M_loc != hi - lo:
        raise ValueError(f"w2_single_sharded: rank {rank} of {world} holds {M_loc} members, its block of n_members={n_members} "
                         f"has {hi - lo} (members are dealt in contiguous blocks, parallel.shard_bounds)")
    d_mu = d_ch = d_w = d_D = None
    try:
        d_mu = gather_member_blocks(ctx, n_members, rank, world, mu_local, (N,))
        d_ch = gather_member_blocks(ctx, n_members, rank, world, chol_local, (N, N))
        d_w, d_D = ctx.empty((n_members,)), ctx.empty((n_members, n_members))
        ctx.w2_single_sharded_dev(n_members, N, d_mu, d_ch, d_w, d_D)
        return d_w.download(), d_D.download()
    finally:
        for d in (d_mu, d_ch, d_w, d_D):
            if d is not None:
                d.free()
