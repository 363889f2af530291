"""world_size-2 rehearsals of the multi-GPU exchange on the CPU.  Each rank owns a block of members,
contributes its [M_loc, P] rows of mean log-likelihood / mean / variance, and `parallel.weighted_posterior_sharded` -
the SAME function bench.py's N > 1 leg calls with the RCCL transport - gathers them and computes the replicated
weights + barycentre.  Transports here: gloo (`HostGather`) and the file rendezvous that bootstraps RCCL
(`FileRendezvous`).  The per-member rows and the normalise / barycentre callbacks come from the oracle (no GPU in
this container)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from bayesian_ensembling_amd import parallel
from oracle import gp_oracle as go


def test_shard_bounds_cover_everything():
    for M in (1, 5, 8, 33, 128):
        for W in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(M, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == M
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            assert max(h - l for l, h in spans) == parallel.padded_rows(M, W)


def test_shard_bounds_match_the_c_abi():
    import ctypes
    from bayesian_ensembling_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)          # pure host arithmetic: callable without a GPU
    lo, hi = ctypes.c_int(), ctypes.c_int()
    for M in (0, 1, 5, 16, 33, 128):
        for W in (1, 2, 3, 8):
            for r in range(W):
                assert lib.bayens_shard_bounds(M, r, W, ctypes.byref(lo), ctypes.byref(hi)) == 0
                assert (lo.value, hi.value) == parallel.shard_bounds(M, r, W)
    assert lib.bayens_shard_bounds(4, 2, 2, ctypes.byref(lo), ctypes.byref(hi)) == -1


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _tables(M, P):
    rng = np.random.default_rng(11)
    return -50.0 - 30.0 * rng.random((M, P)), rng.standard_normal((M, P)), rng.uniform(0.05, 2.0, (M, P))


def _run_rank(rank, world, exchange, M, P, out_dir):
    ll, mean, var = _tables(M, P)
    lo, hi = parallel.shard_bounds(M, rank, world)
    w, mu, v = parallel.weighted_posterior_sharded(
        M, ll[lo:hi], mean[lo:hi], var[lo:hi], exchange,
        lambda l, c: go.loglik_weights(l, c)[0], lambda w_, m_, v_: go.barycentre(w_, m_, v_), c_std=1e-2)
    uid = exchange.broadcast_bytes(bytes(range(128)) if rank == 0 else b"", 128)
    tmax = exchange.max_float(float(rank + 1))
    exchange.barrier()
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), w=w, mu=mu, v=v, uid=np.frombuffer(uid, dtype=np.uint8), tmax=tmax)


def _gloo_worker(rank, world, port, M, P, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _run_rank(rank, world, parallel.HostGather(), M, P, out_dir)
    finally:
        dist.destroy_process_group()


def _file_worker(rank, world, rdzv_dir, M, P, out_dir):
    ex = parallel.FileRendezvous(rdzv_dir, rank, world, timeout_s=60)
    _run_rank(rank, world, ex, M, P, out_dir)
    ex.close()


def _check(tmp_path, M, P, world):
    ll, mean, var = _tables(M, P)
    w_ref, _ = go.loglik_weights(ll, 1e-2)
    mu_ref, v_ref, _ = go.barycentre(w_ref, mean, var)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(got["w"], w_ref) and np.array_equal(got["mu"], mu_ref) and np.array_equal(got["v"], v_ref)
        assert list(got["uid"]) == list(range(128)) and got["tmax"] == float(world)


@pytest.mark.parametrize("M", [5, 8])
def test_two_rank_gather_matches_single_process(tmp_path, M):
    P, world = 37, 2
    mp.spawn(_gloo_worker, args=(world, _free_port(), M, P, str(tmp_path)), nprocs=world, join=True)
    _check(tmp_path, M, P, world)


@pytest.mark.parametrize("M,world", [(5, 2), (7, 3)])
def test_file_rendezvous_gather_matches_single_process(tmp_path, M, world):
    """The transport that bootstraps RCCL (and rehearses N ranks on one GPU), uneven shards included."""
    P = 29
    rdzv = tmp_path / "rdzv"
    mp.spawn(_file_worker, args=(world, str(rdzv), M, P, str(tmp_path)), nprocs=world, join=True)
    _check(tmp_path, M, P, world)
    assert not rdzv.exists()                      # rank 0 removed the rendezvous directory after the last round


def test_w2_single_sharded_carries_uneven_shards():
    """ModelSimilarityWeight(mode='single') sharded over ranks whose shards differ in length (weights.py:240-265 puts no
    condition on the member count): the per-rank blocks go through the fixed-size all-gather padded to ceil(M / world) rows
    and are compacted into the [M][...] table - here with NumPy-backed stand-ins for the device buffers and the collective,
    for 7 members over 3 ranks (blocks 3, 2, 2), 5 over 2, 4 over 4 and the even case 6 over 3."""
    class Buf:
        def __init__(self, shape):
            self.a = np.full(shape, np.nan)
            self.shape = tuple(shape)

        def upload(self, arr):
            flat = np.asarray(arr).reshape(-1)
            self.a.reshape(-1)[:flat.size] = flat
            return self

        def download(self):
            return self.a.copy()

        def copy_rows_from(self, dst_first, src, src_first, count=1):
            self.a[dst_first:dst_first + count] = src.a[src_first:src_first + count]

        def free(self):
            self.a = None

    class FakeCtx:
        """One rank's context; `peers[r]` = what rank r uploads (filled in below, as the collective would see it)."""
        def __init__(self, rank, world, tables):
            self.rank, self.world, self.tables, self.calls = rank, world, tables, 0

        def comm_info(self):
            return self.rank, self.world, "fake"

        def empty(self, shape):
            return Buf(shape)

        def allgather_rows(self, local, all_, rows, width):
            M = sum(len(t[self.calls]) for t in self.tables)
            for r in range(self.world):
                blk = np.full((rows, width), np.nan)
                mine = np.asarray(self.tables[r][self.calls]).reshape(len(self.tables[r][self.calls]), width)
                blk[:len(mine)] = mine
                all_.a.reshape(self.world * rows, width)[r * rows:(r + 1) * rows] = blk
            assert np.array_equal(np.nan_to_num(local.a.reshape(rows, width)[:len(self.tables[self.rank][self.calls])]),
                                  np.nan_to_num(np.asarray(self.tables[self.rank][self.calls]).reshape(-1, width)))
            self.calls += 1

        def w2_single_sharded_dev(self, M, N, mu, chol, w, D):
            self.seen = (mu.a.copy(), chol.a.copy())
            w.a[:] = 1.0 / M
            D.a[:] = 0.0

    rng = np.random.default_rng(3)
    for M, world in ((7, 3), (5, 2), (4, 4), (6, 3), (2, 3)):
        N = 4
        mu, ch = rng.standard_normal((M, N)), rng.standard_normal((M, N, N))
        tables = [(mu[slice(*parallel.shard_bounds(M, r, world))], ch[slice(*parallel.shard_bounds(M, r, world))]) for r in range(world)]
        for r in range(world):
            ctx = FakeCtx(r, world, tables)
            w, D = parallel.w2_single_sharded(ctx, M, tables[r][0], tables[r][1])
            assert np.array_equal(ctx.seen[0], mu) and np.array_equal(ctx.seen[1], ch), (M, world, r)
            assert w.shape == (M,) and D.shape == (M, M)
    with pytest.raises(ValueError, match="holds 3 members"):                 # a rank that holds the wrong block is told so
        parallel.w2_single_sharded(FakeCtx(1, 2, tables), 5, np.zeros((3, 4)), np.zeros((3, 4, 4)))

class _FakeCommCtx:
    """Stands in for _lib.Context in make_exchange: `fail_at` = None | 'load' | 'init' picks where this rank's RCCL breaks;
    `block_init`: comm_init never returns (a rank inside ncclCommInitRank whose peer never joined)."""

    def __init__(self, fail_at, block_init=False):
        self.fail_at, self.block_init, self.inited, self.destroyed, self.ids_drawn = fail_at, block_init, False, False, 0

    def comm_available(self):
        if self.fail_at == "load":
            raise RuntimeError("librccl.so: cannot open shared object file")

    def comm_unique_id(self):
        self.ids_drawn += 1
        return bytes(128)

    def comm_init(self, rank, world, uid):
        if self.fail_at == "init":
            raise RuntimeError("ncclCommInitRank: invalid usage")
        if self.block_init:
            import time
            time.sleep(3600)
        self.inited = True

    def comm_destroy(self):
        self.destroyed = True


def _consensus_worker(rank, world, rdzv_dir, failing_rank, fail_at, out_dir, allow):
    rdzv = parallel.FileRendezvous(rdzv_dir, rank, world, timeout_s=60)
    ctx = _FakeCommCtx(fail_at if rank == failing_rank else None)
    try:
        ex, kind, note = parallel.make_exchange(ctx, rdzv, "rccl", allow_host_exchange=allow)
        res = f"{kind}|{type(ex).__name__}|{ctx.inited}|{ctx.destroyed}|{ctx.ids_drawn}|{note}"
    except parallel.ExchangeUnavailable as e:
        res = f"refused|None|{ctx.inited}|{ctx.destroyed}|{ctx.ids_drawn}|{e}"
    with open(os.path.join(out_dir, f"c{rank}.txt"), "w") as f:
        f.write(res)
    rdzv.close()


@pytest.mark.parametrize("allow", [False, True])
@pytest.mark.parametrize("fail_at,failing_rank", [(None, 0), ("load", 1), ("init", 0), ("init", 2)])
def test_all_ranks_agree_on_the_exchange(tmp_path, fail_at, failing_rank, allow):
    """One rank that cannot load RCCL, or whose communicator does not come up (here: SYMMETRICALLY - every rank's
    comm_init returns), makes EVERY rank refuse (`ExchangeUnavailable`: the strict default) or, only with
    allow_host_exchange, take the host-staged exchange; a rank whose own communicator did come up tears it down again.
    The load probe draws no ncclUniqueId: exactly one id is drawn, by rank 0, and only when every rank can load RCCL."""
    world = 3
    mp.spawn(_consensus_worker, args=(world, str(tmp_path / "rdzv"), failing_rank, fail_at, str(tmp_path), allow),
             nprocs=world, join=True)
    for r in range(world):
        kind, cls, inited, destroyed, ids, note = open(tmp_path / f"c{r}.txt").read().split("|", 5)
        assert int(ids) == (1 if (r == 0 and fail_at != "load") else 0)
        if fail_at is None:
            assert (kind, cls, inited, destroyed, note) == ("rccl", "RcclExchange", "True", "False", "None")
        else:
            if allow:
                assert kind == "file" and cls == "HostStagedExchange" and f"rank {failing_rank}" in note
            else:
                assert kind == "refused" and f"rank {failing_rank}" in note and "--allow-host-exchange" in note
            if fail_at == "load":
                assert inited == "False"                       # nobody entered the communicator
            elif r != failing_rank:
                assert inited == "True" and destroyed == "True"


def _asymmetric_worker(rank, world, rdzv_dir, failing_rank):
    rdzv = parallel.FileRendezvous(rdzv_dir, rank, world, timeout_s=8)
    ctx = _FakeCommCtx("init" if rank == failing_rank else None, block_init=True)
    parallel.make_exchange(ctx, rdzv, "rccl", init_timeout_s=2.0)
    os._exit(0)                                   # not reached by any rank


def test_an_asymmetric_communicator_failure_ends_every_rank_non_zero(tmp_path):
    """ncclCommInitRank failing on ONE rank leaves its peers blocked inside RCCL (the fake comm_init sleeps for an hour):
    the blocked ranks' CommInitWatchdog ends them with exit code 75 after init_timeout_s, the failed rank gives up at
    the rendezvous - nobody hangs, nobody returns 0, and the parent (bench.py spawn_ranks / the launcher) sees a failure."""
    import multiprocessing
    import time
    world, failing = 3, 1
    mpctx = multiprocessing.get_context("spawn")
    procs = [mpctx.Process(target=_asymmetric_worker, args=(r, world, str(tmp_path / "rdzv"), failing)) for r in range(world)]
    t0 = time.time()
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=60)
    assert all(not p.is_alive() for p in procs), "a rank is still blocked inside the (fake) ncclCommInitRank"
    codes = [p.exitcode for p in procs]
    assert codes[0] == codes[2] == parallel.CommInitWatchdog.EXIT_CODE and codes[failing] not in (0, None)
    assert time.time() - t0 < 40


def test_rendezvous_refuses_a_directory_others_can_write_to(tmp_path):
    d = tmp_path / "shared"
    d.mkdir(mode=0o777)
    os.chmod(d, 0o777)
    with pytest.raises(PermissionError):
        parallel.FileRendezvous(str(d), 0, 1)
    os.chmod(d, 0o700)
    parallel.FileRendezvous(str(d), 0, 1).close()


def test_rendezvous_ignores_the_files_of_an_earlier_attempt(tmp_path, monkeypatch):
    """A restarted attempt in the same directory must not read the previous attempt's files: every attempt has its own
    session token, rank 0 clears what earlier attempts left behind, and a non-zero rank takes a session file only if it
    carries the NONCE of its own attempt (not: if the pid in it is alive - a recycled pid, a zombie or a still-running
    earlier rank 0 would all pass that)."""
    d = tmp_path / "rdzv"
    monkeypatch.setenv("BAYENS_RDZV_NONCE", "attempt-1")
    first = parallel.FileRendezvous(str(d), 0, 1)
    assert first.allgather_obj([True, "old"]) == [[True, "old"]]
    stale = sorted(n for n in os.listdir(d) if not n.startswith("session."))
    assert stale
    monkeypatch.setenv("BAYENS_RDZV_NONCE", "attempt-2")
    second = parallel.FileRendezvous(str(d), 0, 1)               # rank 0 of the next attempt: new token, directory cleared
    assert second.session != first.session
    assert not any(name in os.listdir(d) for name in stale)
    assert second.allgather_obj(b"\x01\x02") == [b"\x01\x02"] and second.round == 1
    # a non-zero rank of attempt 3 does not adopt attempt 2's session although the rank 0 that wrote it (this process) is alive
    monkeypatch.setenv("BAYENS_RDZV_NONCE", "attempt-3")
    with pytest.raises(TimeoutError):
        parallel.FileRendezvous(str(d), 1, 2, timeout_s=0.3)
    # ... and takes one that carries its own nonce, whatever pid is in it
    with open(d / "session.0", "wb") as f:
        f.write(parallel._encode({"token": "mine", "pid": 2 ** 22 + 12345, "nonce": "attempt-3"}))
    assert parallel.FileRendezvous(str(d), 1, 2, timeout_s=5).session == "mine"
    # a symlink to a directory this user owns is not a rendezvous directory
    target = tmp_path / "elsewhere"
    target.mkdir(mode=0o700)
    os.symlink(target, tmp_path / "link")
    with pytest.raises(PermissionError):
        parallel.FileRendezvous(str(tmp_path / "link"), 0, 1)
    arr = np.arange(6.0).reshape(2, 3)
    assert np.array_equal(parallel._decode(parallel._encode(arr)), arr) and parallel._decode(parallel._encode(None)) is None
    with pytest.raises(ValueError):
        parallel._decode(b"\x80\x04junk")                        # a pickle stream is not a format this reader accepts


def test_rendezvous_never_touches_files_it_did_not_write(tmp_path, monkeypatch):
    """BAYENS_RDZV may name a directory that already holds a user's files (a results directory): opening a session and closing it
    remove the rendezvous' own names only - session.<n>, <restart>-<hex>.<round>.<rank>, <restart>-<hex>.done.<rank>, their .tmp
    forms - and the directory itself only when nothing else is left in it."""
    d = tmp_path / "results"
    d.mkdir(mode=0o700)
    keep = {"results.json": b"{}", "session_notes.txt": b"mine", "0-notahexname.000001.0": b"x", "run.tmp": b"y"}
    for name, body in keep.items():
        (d / name).write_bytes(body)
    (d / "session.0").write_bytes(b"stale")
    (d / "0-0123456789abcdef.000004.1").write_bytes(b"stale")
    (d / "0-0123456789abcdef.done.1").write_bytes(b"")
    (d / "0-0123456789abcdef.000005.0.tmp").write_bytes(b"stale")
    monkeypatch.setenv("BAYENS_RDZV_NONCE", "attempt-x")
    r = parallel.FileRendezvous(str(d), 0, 1)
    left = set(os.listdir(d))
    assert set(keep) <= left and not any(n.startswith("0-0123456789abcdef") for n in left)
    assert r.allgather_obj("a") == ["a"]
    r.close()
    assert set(os.listdir(d)) == set(keep) and all((d / n).read_bytes() == b for n, b in keep.items())
    # without the torch launcher the fallback nonce does not depend on the parent's pid (ranks under srun / mpirun share none)
    monkeypatch.delenv("BAYENS_RDZV_NONCE")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    assert parallel.FileRendezvous.attempt_nonce().endswith(":-")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job7")
    assert parallel.FileRendezvous.attempt_nonce().endswith(str(os.getppid()))
