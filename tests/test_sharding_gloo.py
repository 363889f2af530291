"""world_size-2 gloo rehearsal of the multi-GPU exchange: each rank owns a block of members,
packs [M_loc, 3, P] rows, all-gathers them and computes the replicated weights + barycentre.  The per-member
rows come from the oracle here (no GPU in this container); on the GPU the same `weighted_posterior_sharded`
runs over ncclAllGather."""
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


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, P, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        ll = -50.0 - 30.0 * rng.random((M, P))
        mean = rng.standard_normal((M, P))
        var = rng.uniform(0.05, 2.0, (M, P))
        lo, hi = parallel.shard_bounds(M, rank, world)
        host = parallel.HostGather()
        w, mu, v = parallel.weighted_posterior_sharded(
            M, ll[lo:hi], mean[lo:hi], var[lo:hi], host,
            lambda l, c: go.loglik_weights(l, c)[0], lambda w_, m_, v_: go.barycentre(w_, m_, v_), c_std=1e-2)
        uid = host.broadcast_bytes(bytes(range(128)) if rank == 0 else b"", 128)
        tmax = host.max_float(float(rank + 1))
        host.barrier()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), w=w, mu=mu, v=v, uid=np.frombuffer(uid, dtype=np.uint8), tmax=tmax)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [5, 8])
def test_two_rank_gather_matches_single_process(tmp_path, M):
    P, world = 37, 2
    mp.spawn(_worker, args=(world, _free_port(), M, P, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(11)
    ll = -50.0 - 30.0 * rng.random((M, P))
    mean = rng.standard_normal((M, P))
    var = rng.uniform(0.05, 2.0, (M, P))
    w_ref, _ = go.loglik_weights(ll, 1e-2)
    mu_ref, v_ref, _ = go.barycentre(w_ref, mean, var)
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(got["w"], w_ref) and np.array_equal(got["mu"], mu_ref) and np.array_equal(got["v"], v_ref)
        assert list(got["uid"]) == list(range(128)) and got["tmax"] == 2.0
