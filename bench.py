#!/usr/bin/env python3
"""bench.py - GP-member-fits/sec + weighted-posterior latency on the 32-member x 2000-timestep synthetic
ensemble (BASELINE.json configs[1]), one process per GPU.

One "step" = one pass of the hot path over the per-rank batch of synthetic members, inputs already resident
in HBM: bayens_gp_fit_batch (Matern-3/2 VGP, n_optim_nits natgrad+Adam iterations, full-cov prediction,
chol/logdet) -> loglik rows -> [RCCL all-gather of the packed (ll, mean, var) rows when N > 1] -> weights
normalisation -> Wasserstein barycentre.  Members shard embarrassingly: each rank fits its own 32 members
("weak" scaling); the only collective is the all-gather of [32, 3, 2000] fp64 rows per rank.

Prints ONE JSON line on rank 0.  Run:  python bench.py --gpus 1 --steps 5 --warmup 1
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet); = 32 FLOP/clk/SIMD x 1024 SIMD x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3  # fp32 matrix peak: the 16x16x4 fp32 MFMA issues in half the cycles of the fp64 one


def fit_flops(N, R, T):
    n3, n2r = float(N) ** 3, 2.0 * N * N * R
    return T * (13.0 / 3.0 * n3 + n2r) + (3.0 * n3 + n2r) + n3 / 3.0


def make_inputs(M, N, R, R_obs, seed):
    """Seeded synthetic ensemble: quadratic trend + AR(1) noise; y_mean = mean over realisations."""
    from bayesian_ensembling_amd.synthetic import synthetic_ensemble
    data, obs = synthetic_ensemble(M, N, R=R, R_obs=R_obs, seed=seed)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
    return X, data.mean(axis=1), data.var(axis=1), obs


def cpu_baseline(N, R, T, members):
    """The oracle (NumPy/LAPACK restatement of the reference arithmetic, validated against the reference's
    pickled fits) timed on this box's host cores, on a bounded sample of the same workload."""
    from threadpoolctl import threadpool_info
    from oracle import gp_oracle as go
    data, _ = go.synthetic_ensemble(members, N, R=R, seed=0xB4E5)
    t0 = time.time()
    for m in range(members):
        X, y, nu = go.prologue(data[m])
        mu, cov, _ = go.gp_fit(X, y, nu, T)
        go.mvn_factor(cov)
    dt = time.time() - t0
    threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    return {"value": members / dt, "unit": "member-fits/s", "cores": int(threads), "kind": "port",
            "sample": f"{members} member(s) of the same workload (N={N}, R={R}, n_optim_nits={T}), "
                      f"NumPy/LAPACK oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=32, help="members per GPU")
    ap.add_argument("--timesteps", type=int, default=2000)
    ap.add_argument("--realisations", type=int, default=8)
    ap.add_argument("--nits", type=int, default=2, help="n_optim_nits (2 = the reference's test scale)")
    ap.add_argument("--kernel", default="matern32", choices=["matern32", "matern52", "rbf"])
    ap.add_argument("--groups", type=int, default=1, help="member groups = HIP streams per fit batch")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"],
                    help="f64 = the reference precision (the headline); f32 = BASELINE config 3 (1e-3 parity bar)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-members", type=int, default=2)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    force_comm = bool(os.environ.get("BAYENS_FORCE_COMM"))    # exercise the RCCL path with a single rank
    if world > 1 or force_comm:
        # torch (gloo) is the control plane of the multi-process run.  It must be imported BEFORE libbayens: torch
        # bundles its own libamdhip64 / librccl (same sonames as /opt/rocm's); whichever HIP runtime is loaded
        # first is the one the whole process uses, and RCCL only initialises on the runtime it was built against.
        import torch.distributed as dist
    from bayesian_ensembling_amd import _lib, parallel
    n_dev = _lib.device_count()
    if n_dev <= 0:
        raise SystemExit("no HIP device visible: the hot path needs an MI355X (there is no CPU fallback)")
    ctx = _lib.Context(local_rank % n_dev)  # (a launcher may expose a single device to every rank)
    ctx.set_groups(args.groups)
    host = None
    if world > 1 or force_comm:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only (barrier, max, uid)
        host = parallel.HostGather()
        parallel.init_rccl(ctx, host)       # data plane: ncclAllGather over xGMI inside libbayens

    M, N, R, T, R_obs = args.members, args.timesteps, args.realisations, args.nits, 3
    kid = _lib.KERNEL_IDS[args.kernel]
    X, y, nu, obs = make_inputs(M, N, R, R_obs, seed=0xB4E5 + rank)
    f32 = args.dtype == "f32"
    dt = np.float32 if f32 else np.float64
    dX, dy, dnu, dobs = ctx.to_device(X, dt), ctx.to_device(y, dt), ctx.to_device(nu, dt), ctx.to_device(obs)
    dmu, dcov, dchol = ctx.empty((M, N), dt), ctx.empty((M, N, N), dt), ctx.empty((M, N, N), dt)
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N), dt)
    if f32:    # the weight / scheme kernels take fp64 tables: the fp32 posterior's [M][N] rows are widened (a few MB)
        dmu64, dbasis64 = ctx.empty((M, N)), ctx.empty((M, 2, N))
    else:
        dmu64, dbasis64 = dmu, dbasis
    Mtot = M * world
    dll_loc, dvar_loc = ctx.empty((M, N)), ctx.empty((M, N))
    if host:
        dll_all, dmean_all, dvar_all = ctx.empty((Mtot, N)), ctx.empty((Mtot, N)), ctx.empty((Mtot, N))
    else:
        dll_all, dmean_all, dvar_all = dll_loc, dmu64, dvar_loc
    dw, dbmu, dbvar = ctx.empty((Mtot, N)), ctx.empty((N,)), ctx.empty((N,))

    def fit():
        return ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, T, dmu, dcov, dchol, dld, dhyp, kernel=kid, basis=dbasis,
                                    f32=f32)

    def weighted_posterior():
        # LogLikelihoodWeight rows + posterior marginals of the local members, exchange of the [M, N] rows over
        # RCCL (log-likelihoods, means, variances), then normalisation over ALL members and the barycentre
        if not host and not f32:
            # one GPU: the whole weighted posterior (rows -> weights -> barycentre) is one entry point, one host wait
            return ctx.weighted_posterior_dev(M, N, R_obs, dbasis, dld, dobs, dmu, dcov, dw, dbmu, dbvar, c_std=1e-3)
        if f32:
            ctx.widen_f32_dev(M * 2 * N, dbasis, dbasis64)
            ctx.widen_f32_dev(M * N, dmu, dmu64)
        ctx.loglik_rows_basis_dev(M, N, R_obs, dbasis64, dld, dobs, dll_loc)
        ctx.extract_diag_dev(M, N, dcov, dvar_loc, f32=f32)
        if host:
            ctx.allgather_rows(dll_loc, dll_all, M, N)
            ctx.allgather_rows(dmu64, dmean_all, M, N)
            ctx.allgather_rows(dvar_loc, dvar_all, M, N)
        ctx.weights_normalise_dev(Mtot, N, dll_all, 1e-3, dw)
        return ctx.barycentre_dev(Mtot, N, dw, dmean_all, dvar_all, dbmu, dbvar)

    def step():
        fit()
        return weighted_posterior()

    def barrier():
        ctx.sync()
        if host:
            host.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.sync()
    barrier()
    elapsed = time.perf_counter() - t0
    if host:
        elapsed = host.max_float(elapsed)
    ms_per_step = elapsed / args.steps * 1e3
    value = Mtot * args.steps / elapsed

    # ---- untimed diagnostics on rank 0: latencies with HIP events, GEMM roofline, CPU baseline ----
    extra = {}
    roof = None
    cpu = None
    if rank == 0:
        ctx.timer_start(); fit(); fit_ms = ctx.timer_stop()
        ctx.timer_start(); bad = weighted_posterior(); wp_ms = ctx.timer_stop()
        # the stand-alone entry point (mu, chol, logdet -> rows: two triangular solves per member) for comparison
        ll_chol_ms = None
        if not f32:
            ctx.timer_start(); ctx.loglik_rows_dev(M, N, R_obs, dmu, dchol, dld, dobs, dw); ll_chol_ms = ctx.timer_stop()
        ctx.set_groups(1)                 # per-kernel durations are only meaningful when kernels do not overlap
        fit()
        ctx.timer_start(); fit(); fit_serial_ms = ctx.timer_stop()
        ctx.profile_enable(True)
        fit()
        p = ctx.profile_get(1)            # the dominant kernel: gemm_f64_mfma<false,true> (A row-major, B^T)
        p_all = ctx.profile_get(0)
        stages = ctx.profile_stages()
        if os.environ.get("BAYENS_PRINT_SHAPES"):
            print(ctx.profile_shapes(), file=sys.stderr, flush=True)
        ctx.profile_enable(False)
        if os.environ.get("BAYENS_PRINT_STAGES"):
            print(stages, file=sys.stderr, flush=True)
        ctx.set_groups(args.groups)
        gemm_tflops = p["flop"] / (p["ms"] * 1e-3) / 1e12
        traffic = None     # HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes
        peak = F32_MFMA_PEAK_TFLOPS if f32 else F64_MFMA_PEAK_TFLOPS
        kname = "bayens_f32::gemm_f32_mfma" if f32 else "bayens::gemm_f64_mfma"
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if files and not f32:    # (the PMC passes were collected on the fp64 instance)
                traffic = round(json.load(open(files[-1]))["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
        group_tflops = p_all["flop"] / (p_all["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": kname + "<false, true>", "achieved": round(gemm_tflops, 3),
                "peak": peak, "unit": "TFLOP/s", "frac": round(gemm_tflops / peak, 4),
                "traffic": traffic, "algorithmic_bytes_per_launch": round(p["bytes"] / max(p["launches"], 1)),
                "launches_per_fit_batch": p["launches"],
                "avg_launch_us": round(p["ms"] * 1e3 / max(p["launches"], 1), 2),
                "kernel_ms_per_fit_batch": round(p["ms"], 3),
                "gemm_group": {"kernels": kname.split("::")[1] + "<*>, gemm_panel64, trsm_panel_kernel",
                               "achieved": round(group_tflops, 3), "frac": round(group_tflops / peak, 4),
                               "launches": p_all["launches"], "ms_per_fit_batch": round(p_all["ms"], 3)},
                "fit_batch_ms_one_stream": round(fit_serial_ms, 3),
                "note": "HIP events around every launch of the kernel in one fit batch, one stream (no overlap); "
                        "traffic = HBM bytes per launch of this kernel from the rocprofv3 --pmc passes in profiles/*_pmc_traffic.json"}
        extra = {"fit_batch_ms": round(fit_ms, 3), "weighted_posterior_latency_ms": round(wp_ms, 4),
                 "fit_algorithmic_tflops": round(M * fit_flops(N, R, T) / (fit_ms * 1e-3) / 1e12, 3),
                 "fit_frac_of_mfma_peak": round(M * fit_flops(N, R, T) / (fit_ms * 1e-3) / 1e12 / peak, 4),
                 "barycentre_not_converged": int(bad),
                 "loglik_rows_from_chol_ms": None if ll_chol_ms is None else round(ll_chol_ms, 4)}
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(N, R, T, args.cpu_members)
    if host:
        host.barrier()
        ctx.comm_destroy()
        dist.destroy_process_group()

    if rank == 0:
        out = {
            "metric": "GP-member-fits/sec + weighted-posterior latency, 32-member x 2000-step ensemble",
            "value": round(value, 3), "unit": "member-fits/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{M}-member x {N}-timestep synthetic ensemble per GPU, R={R} realisations, "
                                   f"{args.kernel} VGP fit (n_optim_nits={T}) + LogLikelihoodWeight(c=1e-3) + Barycentre",
                       "members_per_gpu": M, "timesteps": N, "realisations": R, "n_optim_nits": T,
                       "kernel": args.kernel, "obs_realisations": R_obs, "member_groups": args.groups,
                       "parallelism": f"member-sharded x{world}, RCCL all-gather of [M,3,N] rows" if world > 1 else "single GPU"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        out.update(extra)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
