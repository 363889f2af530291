#!/usr/bin/env python3
"""bench.py - GP-member-fits/sec + weighted-posterior latency on the 32-member x 2000-timestep synthetic
ensemble (BASELINE.json configs[1]), one process per GPU.

One "step" = one pass of the hot path over the per-rank batch of synthetic members, inputs already resident
in HBM: bayens_gp_fit_batch (Matern-3/2 VGP, n_optim_nits natgrad+Adam iterations, full-cov prediction,
chol/logdet) -> loglik rows -> [RCCL all-gather of the packed (ll, mean, var) rows when N > 1] -> weights
normalisation -> Wasserstein barycentre.  Members shard embarrassingly: each rank fits its own members ("weak"
scaling); the only collective is ONE all-gather of [M_loc, 3, 2000] fp64 rows per rank
(bayesian_ensembling_amd.parallel.weighted_posterior_sharded - the function the CPU gloo test drives too).

Launch:  python bench.py --gpus N --steps K --warmup W
  * under a launcher (RANK / LOCAL_RANK / WORLD_SIZE set, e.g. python -m torch.distributed.run): this process is one rank;
  * plainly with --gpus N > 1: N child ranks are started here, BEFORE anything touches a GPU, one device each.
No PyTorch in the ranks: the 128-byte RCCL id travels through a file rendezvous, barrier and max-over-ranks are an
ncclAllReduce.  Rank 0 prints ONE JSON line.

Presets: --config c2 (default: 32 members/GPU, fp64 Matern-3/2), c3 (fp32 Matern-5/2, weights = ModelSimilarityWeight
mode='temporal'; its mode='single' table is timed beside it in the default line), c4 (16 members/GPU = 128 members
over 8 GPUs).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak (datasheet); = 32 FLOP/clk/SIMD x 1024 SIMD x 2.4 GHz
F32_MFMA_PEAK_TFLOPS = 157.3  # fp32 matrix peak (MI355X_MICROARCH.md: 16x16x4 fp32 MFMA, 64 FLOP/clk/SIMD)
METRIC = "GP-member-fits/sec + weighted-posterior latency, 32-member x 2000-step ensemble"


def fit_flops(N, R, T, f32=False):
    """Algorithmic flop of one member fit (= fit_flops in csrc/driver.cpp): 13/3 N^3 per iteration, 3 N^3 in
    a lone first one (Lam_old = I: G S = (I - S) / gamma needs S only; fp64 only - fp32 keeps the general form there) and
    5/3 N^3 in a first one that another follows (its hyper-gradient from P = (diag(nu) + gamma K)^-1 alone), 3 N^3 for the
    prediction, N^3/3 for chol(cov).  The count is the fit WITHOUT loss samples (a sampled run executes more; no line that
    quotes a flop rate requests samples)."""
    n3, n2r = float(N) ** 3, 2.0 * N * N * R
    first_saves = 8.0 / 3.0 * n3 if T >= 2 else (4.0 / 3.0 * n3 if T == 1 and not f32 else 0.0)
    return T * (13.0 / 3.0 * n3 + n2r) - first_saves + (3.0 * n3 + n2r) + n3 / 3.0


def check_known_answer(name, hyp0, mu0, trace_cov):
    """A long timed run held to the oracle's stored answer for member 0 (tests/golden/bench_values.json, written by
    oracle/make_bench_values.py from the NumPy oracle): final (variance, lengthscale), mean(mu), trace(cov) within the 1e-5
    relative bar BASELINE.json states for fp64.  Returns the worst relative deviation; raises when outside the bar."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "bench_values.json")
    ref = json.load(open(path))[name]
    dev = max(abs(float(hyp0[0]) / ref["variance"] - 1.0), abs(float(hyp0[1]) / ref["lengthscale"] - 1.0),
              abs(float(np.mean(mu0)) / ref["mean_mu"] - 1.0), abs(float(trace_cov) / ref["trace_cov"] - 1.0))
    if not (np.isfinite(dev) and dev < 1e-5):
        raise AssertionError(f"{name}: member 0 is {dev:.2e} (relative) away from the oracle's stored answer")
    return dev


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


class Workload:
    """Resident inputs/outputs of one configuration on one rank and the step over them."""

    def __init__(self, ctx, M, N, R, T, kernel, f32, rank, world, exchange, groups=1, R_obs=3, weighting="loglik"):
        from bayesian_ensembling_amd import _lib, parallel
        self.ctx, self.M, self.N, self.R, self.T, self.f32, self.R_obs = ctx, M, N, R, T, f32, R_obs
        self.kernel, self.kid, self.world, self.exchange, self.parallel = kernel, _lib.KERNEL_IDS[kernel], world, exchange, parallel
        self.Mtot = M * world
        self.weighting = weighting        # "loglik": LogLikelihoodWeight; "w2_temporal": ModelSimilarityWeight(mode='temporal')
        X, y, nu, obs = make_inputs(M, N, R, R_obs, seed=0xB4E5 + rank)
        dt = np.float32 if f32 else np.float64
        self.dX, self.dy, self.dnu, self.dobs = ctx.to_device(X, dt), ctx.to_device(y, dt), ctx.to_device(nu, dt), ctx.to_device(obs)
        self.dmu, self.dcov, self.dchol = ctx.empty((M, N), dt), ctx.empty((M, N, N), dt), ctx.empty((M, N, N), dt)
        self.dld, self.dhyp, self.dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N), dt)
        if f32:   # the weight / scheme kernels take fp64 tables: the fp32 posterior's [M][N] rows are widened (a few MB)
            self.dmu64, self.dbasis64 = ctx.empty((M, N)), ctx.empty((M, 2, N))
        else:
            self.dmu64, self.dbasis64 = self.dmu, self.dbasis
        self.dll_loc, self.dvar_loc = ctx.empty((M, N)), ctx.empty((M, N))
        self.dw, self.dbmu, self.dbvar = ctx.empty((self.Mtot, N)), ctx.empty((N,)), ctx.empty((N,))
        self.bufs = [self.dX, self.dy, self.dnu, self.dobs, self.dmu, self.dcov, self.dchol, self.dld, self.dhyp, self.dbasis,
                     self.dll_loc, self.dvar_loc, self.dw, self.dbmu, self.dbvar] + ([self.dmu64, self.dbasis64] if f32 else [])

    def free(self):
        for b in self.bufs:
            b.free()

    def fit(self, T=None):
        return self.ctx.gp_fit_batch_dev(self.M, self.N, self.R, self.dX, self.dy, self.dnu, self.T if T is None else T,
                                         self.dmu, self.dcov, self.dchol, self.dld, self.dhyp, kernel=self.kid,
                                         basis=self.dbasis, f32=self.f32)

    def local_rows(self):
        """LogLikelihoodWeight rows + posterior marginals of this rank's members ([M, N] each, resident)."""
        c = self.ctx
        if self.f32:
            c.widen_f32_dev(self.M * 2 * self.N, self.dbasis, self.dbasis64)
            c.widen_f32_dev(self.M * self.N, self.dmu, self.dmu64)
        c.loglik_rows_basis_dev(self.M, self.N, self.R_obs, self.dbasis64, self.dld, self.dobs, self.dll_loc)
        c.extract_diag_dev(self.M, self.N, self.dcov, self.dvar_loc, f32=self.f32)

    def weighted_posterior(self, local_only=False):
        """Weights over ALL members + barycentre.  local_only: this rank's members alone, no collective (the untimed
        rank-0 diagnostics - a collective there would wait for ranks that have already left)."""
        c, M, N = self.ctx, self.M, self.N
        temporal = self.weighting == "w2_temporal"
        if (self.exchange is None or local_only) and not self.f32 and not temporal:
            # one GPU: the whole weighted posterior (rows -> weights -> barycentre) is one entry point, one host wait
            return c.weighted_posterior_dev(M, N, self.R_obs, self.dbasis, self.dld, self.dobs, self.dmu, self.dcov, self.dw,
                                            self.dbmu, self.dbvar, c_std=1e-3)
        self.local_rows()
        if self.exchange is None or local_only:
            if temporal:
                c.w2_temporal_dev(M, N, self.dmu64, self.dvar_loc, self.dw)
            else:
                c.weights_normalise_dev(M, N, self.dll_loc, 1e-3, self.dw)
            return c.barycentre_dev(M, N, self.dw, self.dmu64, self.dvar_loc, self.dbmu, self.dbvar)
        bad = []

        def normalise(ll, c_std):
            c.weights_normalise_dev(self.Mtot, N, ll, c_std, self.dw)
            return self.dw

        def barycentre(w, mean, var):
            bad.append(c.barycentre_dev(self.Mtot, N, w, mean, var, self.dbmu, self.dbvar))
            return self.dbmu, self.dbvar, bad[-1]

        def similarity(ll, mean, var):
            c.w2_temporal_dev(self.Mtot, N, mean, var, self.dw)
            return self.dw

        self.parallel.weighted_posterior_sharded(self.Mtot, self.dll_loc, self.dmu64, self.dvar_loc, self.exchange,
                                                 normalise, barycentre, c_std=1e-3, weights=similarity if temporal else None)
        return bad[-1]

    def step(self):
        self.fit()
        return self.weighted_posterior()

    def timed(self, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + device sync on both sides; max over ranks."""
        for _ in range(warmup):
            self.step()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        elapsed = time.perf_counter() - t0
        if self.exchange is not None:
            elapsed = self.exchange.max_float(elapsed)
        return elapsed

    def barrier(self):
        self.ctx.sync()
        if self.exchange is not None:
            self.exchange.barrier()

    def roofline(self):
        """Per-launch HIP events around every launch of the GEMM group in ONE fit batch, on the stream they run on
        (look-ahead off so that kernels do not overlap).  The dominant kernel is the NT GEMM."""
        c, f32, N = self.ctx, self.f32, self.N
        c.set_groups(1)
        self.fit()
        c.timer_start(); self.fit(); fit_serial_ms = c.timer_stop()
        c.profile_enable(True)
        self.fit()
        p, p_all = c.profile_get(1), c.profile_get(0)
        small_tiles = p["launches"] == 0        # a small shape: every NT product took the 64 x 64-tile kernel (family 6)
        if small_tiles:
            p = c.profile_get(6)
        stages = c.profile_stages()
        if os.environ.get("BAYENS_PRINT_SHAPES"):
            print(c.profile_shapes(), file=sys.stderr, flush=True)
        c.profile_enable(False)
        if os.environ.get("BAYENS_PRINT_STAGES"):
            print(stages, file=sys.stderr, flush=True)
        peak = F32_MFMA_PEAK_TFLOPS if f32 else F64_MFMA_PEAK_TFLOPS
        kname = "bayens_f32::gemm_f32_mfma" if f32 else "bayens::gemm_f64_mfma"
        # the launches are issued on members padded to Np = ceil(N/64)*64: the flop the kernel is credited with here is
        # the ALGORITHMIC flop of the unpadded problem, (N/Np)^3 of what the padded launches compute
        Np = -(-N // 64) * 64
        unpad = (N / Np) ** 3
        tfl_padded = p["flop"] / (p["ms"] * 1e-3) / 1e12
        tfl = tfl_padded * unpad
        group = p_all["flop"] / (p_all["ms"] * 1e-3) / 1e12 * unpad
        traffic, traffic_from = None, None
        try:
            import glob
            files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
            if files and not f32 and (self.M, N, self.T) == (32, 2000, 2):    # the PMC passes were collected on this workload
                traffic = round(json.load(open(files[-1]))["hbm_bytes_per_launch"])
                traffic_from = os.path.basename(files[-1])
        except Exception:
            traffic = None
        if small_tiles:
            kname = kname.replace("_mfma", "").replace("gemm_f64", "gemm_tile64").replace("gemm_f32", "gemm_tile64")
        return {"bound": "mfma", "kernel": kname + ("<*>" if small_tiles else "<false, true>"), "achieved": round(tfl, 3), "peak": peak,
                "unit": "TFLOP/s", "frac": round(tfl / peak, 4), "traffic": traffic, "traffic_source": traffic_from,
                "achieved_on_padded_dims": round(tfl_padded, 3), "frac_on_padded_dims": round(tfl_padded / peak, 4),
                "padded_order": Np,
                "algorithmic_bytes_per_launch": round(p["bytes"] / max(p["launches"], 1)),
                "launches_per_fit_batch": p["launches"],
                "avg_launch_us": round(p["ms"] * 1e3 / max(p["launches"], 1), 2),
                "kernel_ms_per_fit_batch": round(p["ms"], 3),
                "gemm_group": {"kernels": kname.split("::")[1] + "<*>, gemm_tile64, gemm_panel64, trsm_panel_kernel",
                               "achieved": round(group, 3), "frac": round(group / peak, 4),
                               "launches": p_all["launches"], "ms_per_fit_batch": round(p_all["ms"], 3)},
                "fit_batch_ms_one_stream": round(fit_serial_ms, 3),
                "note": "HIP events around every launch of the kernel in one fit batch, look-ahead stream off (kernels strictly one after the "
                        "other; the timed step runs with the look-ahead, whose kernels overlap and cannot be priced one by one); achieved = "
                        "algorithmic flop of the UNPADDED problem / summed launch time; traffic = HBM bytes per launch of "
                        "this kernel from the rocprofv3 --pmc passes named in traffic_source (not measured in this run)"}


HBM_PEAK_GBS = 8000.0      # MI355X HBM3E (MI355X_MICROARCH.md: 8.0 TB/s spec; 6.29 TB/s measured copy)


class FieldWorkload:
    """BASELINE.json config 5: one independent GP per (member, grid point) - M = 32 members x G grid points per GPU
    (96 x 144 = 13 824 points over 8 GPUs = 1 728 per GPU), N = 86 annual steps, R = 8.  One step = ONE launch of the
    fused small-member kernel over the B = M * G fits (reduced export: mu, diag(cov), basis, logdet; or the full
    covariance) -> per-gridpoint LogLikelihoodWeight rows -> normalisation over the members -> barycentre per point."""

    def __init__(self, ctx, M, G, N, R, T, export, R_obs=3, seed=0xC5, f32=False, keep_host=False):
        self.ctx, self.M, self.G, self.N, self.R, self.T, self.export, self.R_obs = ctx, M, G, N, R, T, export, R_obs
        self.f32 = f32
        if f32:
            export = self.export = "full"          # the fp32 entry point exports the covariance (no reduced fp32 entry)
        B = self.B = M * G
        rng = np.random.default_rng(seed)
        tau = np.linspace(0.0, 1.0, N)
        amp = 0.8 + 0.4 * rng.random((M, 1, 1, 1))                      # member signal, shared by its grid points
        loc = 1.0 + 0.2 * rng.standard_normal((1, G, 1, 1))             # grid-point offset of the trend
        data = 1.5 * amp * loc * tau ** 2 + 0.036 * rng.standard_normal((M, G, R, N)).cumsum(axis=3)
        obs = 1.5 * loc[0] * tau ** 2 + 0.036 * rng.standard_normal((G, R_obs, N)).cumsum(axis=2)
        data = data.reshape(B, R, N)
        X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
        dt = np.float32 if f32 else np.float64
        self.dX, self.dy, self.dnu = ctx.to_device(X, dt), ctx.to_device(data.mean(axis=1), dt), ctx.to_device(data.var(axis=1), dt)
        self.dobs = ctx.to_device(obs)
        # (tests/test_gpu_small.py holds THIS launch against the oracle: it asks for the host copies of the inputs)
        self.host = (X, data.mean(axis=1), data.var(axis=1), obs) if keep_host else None
        self.dmu, self.dvar, self.dld, self.dhyp = ctx.empty((B, N), dt), ctx.empty((B, N)), ctx.empty((B,)), ctx.empty((B, 2))
        self.dbasis, self.dll, self.dw = ctx.empty((B, 2, N), dt), ctx.empty((B, N)), ctx.empty((B, N))
        self.dcov = ctx.empty((B, N, N), dt) if export == "full" else None
        self.dbmu, self.dbvar = ctx.empty((G * N,)), ctx.empty((G * N,))
        self.dmu64, self.dbasis64 = (ctx.empty((B, N)), ctx.empty((B, 2, N))) if f32 else (self.dmu, self.dbasis)
        self.bufs = [b for b in (self.dX, self.dy, self.dnu, self.dobs, self.dmu, self.dvar, self.dld, self.dhyp, self.dbasis,
                                 self.dll, self.dw, self.dcov, self.dbmu, self.dbvar) if b is not None]
        if f32:
            self.bufs += [self.dmu64, self.dbasis64]

    def free(self):
        for b in self.bufs:
            b.free()

    def fit(self):
        c = self.ctx
        if self.export == "full":
            info = c.gp_fit_batch_dev(self.B, self.N, self.R, self.dX, self.dy, self.dnu, self.T, self.dmu, self.dcov, None,
                                      self.dld, self.dhyp, basis=self.dbasis, f32=self.f32)
            c.extract_diag_dev(self.B, self.N, self.dcov, self.dvar, f32=self.f32)
            if self.f32:
                c.widen_f32_dev(self.B * self.N, self.dmu, self.dmu64)
                c.widen_f32_dev(self.B * 2 * self.N, self.dbasis, self.dbasis64)
            return info
        return c.gp_fit_batch_reduced_dev(self.B, self.N, self.R, self.dX, self.dy, self.dnu, self.T, self.dmu, self.dvar,
                                          self.dld, self.dhyp, self.dbasis)

    def weighted_posterior(self):
        c, P = self.ctx, self.G * self.N
        c.loglik_rows_basis_grouped(self.dbasis64, self.dld, self.dobs, loc=1, out=self.dll)
        c.weights_normalise_dev(self.M, P, self.dll, 1e-3, self.dw)
        return c.barycentre_dev(self.M, P, self.dw, self.dmu64, self.dvar, self.dbmu, self.dbvar)

    def step(self):
        self.fit()
        return self.weighted_posterior()

    def measure(self, steps, warmup):
        c = self.ctx
        for _ in range(warmup):
            self.step()
        c.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        c.sync()
        elapsed = time.perf_counter() - t0
        c.timer_start(); self.fit(); fit_ms = c.timer_stop()          # the dominant kernel: HIP events on its stream
        c.timer_start(); bad = self.weighted_posterior(); wp_ms = c.timer_stop()
        N, R = self.N, self.R
        es = 4.0 if self.f32 else 8.0
        in_b = es * (N * R + 2 * N)
        out_b = es * (N * N + N) if self.export == "full" else 8.0 * (4 * N + 3)
        gbs = self.B * (in_b + out_b) / (fit_ms * 1e-3) / 1e9
        tfl = self.B * fit_flops(N, R, self.T, self.f32) / (fit_ms * 1e-3) / 1e12
        peak_tfl = F32_MFMA_PEAK_TFLOPS if self.f32 else F64_MFMA_PEAK_TFLOPS
        return {"value": round(self.B * steps / elapsed, 1), "unit": "member-fits/s", "ms_per_step": round(elapsed / steps * 1e3, 3),
                "fits_per_step": self.B, "export": self.export, "fit_kernel_ms": round(fit_ms, 3),
                "weighted_posterior_latency_ms": round(wp_ms, 4), "barycentre_not_converged": int(bad),
                # primary roofline of config 5: the fp64 matrix pipes (SURVEY 8(d)'s HBM floor for this config was
                # mis-scaled by 10^3 - 29.5 GB at 6.29 TB/s is 4.7 ms, not 4.7 s - so HBM does not bind; the HBM figure stays beside it)
                "roofline": {"bound": "mfma", "kernel": ("bayens_f32" if self.f32 else "bayens") + "::smalln::small_fit_kernel<6>",
                             "achieved": round(tfl, 3), "peak": peak_tfl, "unit": "TFLOP/s", "frac": round(tfl / peak_tfl, 4), "traffic": None,
                             "algorithmic_flop_per_fit": fit_flops(N, R, self.T), "launches_per_step": 1,
                             "avg_launch_us": round(fit_ms * 1e3, 1),
                             "note": "one launch of the fused small-member kernel per step; achieved = algorithmic flop of the B fits / its "
                                     "duration (HIP events on its stream).  SURVEY 8(d) priced config 5 against HBM with a floor that "
                                     "was mis-scaled by 10^3: with every matrix resident in LDS the kernel moves 9.7 KB per fit and is "
                                     "bound by the fp64 matrix / vector pipes and its serial 16x16 factorisations - see roofline_hbm"},
                "roofline_hbm": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_fit": in_b + out_b},
                "roofline_mfma": {"bound": "mfma", "achieved": round(tfl, 3), "peak": peak_tfl, "unit": "TFLOP/s",
                                  "frac": round(tfl / peak_tfl, 4), "algorithmic_flop_per_fit": fit_flops(N, R, self.T)}}

def cpu_baseline_small(N, R, T, fits):
    from threadpoolctl import threadpool_info
    from oracle import gp_oracle as go
    data, _ = go.synthetic_ensemble(fits, N, R=R, seed=0xC5)
    t0 = time.time()
    for m in range(fits):
        X, y, nu = go.prologue(data[m])
        mu, cov, _ = go.gp_fit(X, y, nu, T)
        go.mvn_factor(cov)
    dt = time.time() - t0
    threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    return {"value": fits / dt, "unit": "member-fits/s", "cores": int(threads), "kind": "port",
            "sample": f"{fits} fits of the same shape (N={N}, R={R}, n_optim_nits={T}), NumPy/LAPACK oracle, {dt:.1f} s"}


def api_path(ctx, M, N, R, T, reps=5):
    """The same step through the plug-in classes a user of the reference calls: ModelCollection.fit(GPDTW1D) ->
    LogLikelihoodWeight -> Barycentre (ensembles/data.py:385-395, weights.py:87-131, ensemble_scheme.py:43-81), host
    arrays in, posteriors resident in HBM (resident.py).  Includes the PCIe upload of the realisations, the Python
    containers and the download of the [M, N] weight table and the barycentre."""
    import warnings
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    from bayesian_ensembling_amd.synthetic import synthetic_ensemble
    data, obs = synthetic_ensemble(M, N, R=R, R_obs=3, seed=0xB4E5)
    tcoord = np.arange(N)
    mk = lambda a: DataArray(a, ("realisation", "time"), {"realisation": np.arange(a.shape[0]), "time": tcoord})
    mc = es.ModelCollection([es.ProcessModel(mk(data[m]), f"member{m}") for m in range(M)])
    ob = es.ProcessModel(mk(obs), "obs")
    times = {}
    for bar in (False, True):
        best = None
        for _ in range(reps + 1):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                t0 = time.perf_counter()
                mc.fit(model=es.GPDTW1D(), n_optim_nits=T, progress_bar=bar)
                w = es.LogLikelihoodWeight()(mc, ob, standardisation_constant=1e-3)
                post = es.Barycentre()(mc, w)
                dt = time.perf_counter() - t0
            if os.environ.get("BAYENS_PRINT_STAGES"):
                print(f"api_path progress_bar={bar}: {dt * 1e3:.1f} ms", file=sys.stderr, flush=True)
            best = dt if best is None else min(best, dt)
        times[bar] = best
    assert np.isfinite(post.mu).all()
    return {"api_path_fits_per_s": round(M / times[False], 2), "api_path_ms_per_step": round(times[False] * 1e3, 2),
            "api_path_fits_per_s_with_loss_samples": round(M / times[True], 2),
            "api_path_note": "ModelCollection.fit(GPDTW1D, progress_bar=False) + LogLikelihoodWeight + Barycentre through the "
                             "Python classes, best of %d; *_with_loss_samples = progress_bar=True (the reference default: one "
                             "-ELBO sample per 25 iterations)" % reps}


def api_path_c5(ctx, M, n_lat, n_lon, N, R, T, reps=3):
    """BASELINE config 5 through the plug-in classes: ModelCollection.fit(GPDTW1D(per_gridpoint=True)) ->
    LogLikelihoodWeight -> Barycentre on (realisation, time, latitude, longitude) models, host arrays in (the realisations
    cross PCIe as they lie in memory; the prologue runs on the device), per-gridpoint posteriors resident in HBM, the
    [M, time, lat, lon] weight table and the barycentre field downloaded."""
    import warnings
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    rng = np.random.default_rng(0xC5)
    G = n_lat * n_lon
    tau = np.linspace(0.0, 1.0, N)[None, :, None, None]
    loc = 1.0 + 0.2 * rng.standard_normal((1, 1, n_lat, n_lon))
    coords = {"time": np.arange(N), "latitude": np.linspace(-89, 89, n_lat), "longitude": np.linspace(0, 357.5, n_lon)}
    dims = ("realisation", "time", "latitude", "longitude")

    def pm(amp, n_real, name):
        vals = 1.5 * amp * loc * tau ** 2 + 0.036 * rng.standard_normal((n_real, N, n_lat, n_lon)).cumsum(axis=1)
        return es.ProcessModel(DataArray(vals, dims, dict(coords, realisation=np.arange(n_real))), name)
    mc = es.ModelCollection([pm(0.8 + 0.4 * rng.random(), R, f"member{m}") for m in range(M)])
    ob = pm(1.0, 3, "obs")
    best, parts = None, None
    for _ in range(reps + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            t0 = time.perf_counter()
            mc.fit(model=es.GPDTW1D(per_gridpoint=True), n_optim_nits=T, progress_bar=False)
            ctx.sync()
            t1 = time.perf_counter()
            w = es.LogLikelihoodWeight()(mc, ob, standardisation_constant=1e-3)
            t2 = time.perf_counter()
            post = es.Barycentre()(mc, w)
            t3 = time.perf_counter()
        if best is None or t3 - t0 < best:
            best, parts = t3 - t0, (t1 - t0, t2 - t1, t3 - t2)
    t4 = time.perf_counter()
    wv = np.asarray(w.values)                 # the [M, time, lat, lon] table stays in HBM until somebody reads it: timed apart
    w_dl = time.perf_counter() - t4
    assert np.isfinite(post.mu).all() and w.shape == (M, N, n_lat, n_lon) and np.isfinite(wv).all()
    return {"api_path_c5_fits_per_s": round(M * G / best, 1), "api_path_c5_ms_per_step": round(best * 1e3, 2),
            "api_path_c5_ms_fit_weights_barycentre": [round(p * 1e3, 2) for p in parts],
            "api_path_c5_weight_table_download_ms": round(w_dl * 1e3, 2),
            "api_path_c5_note": f"{M} members x {n_lat}x{n_lon} grid points, N={N}, R={R}: ModelCollection.fit(GPDTW1D("
                                "per_gridpoint=True)) + LogLikelihoodWeight + Barycentre through the Python classes, host "
                                f"realisations in ({M * R * N * G * 8 / 1e6:.0f} MB over PCIe), barycentre field out; the "
                                f"{M * N * G * 8 / 1e6:.0f} MB weight table stays resident for the scheme and is downloaded when a "
                                "caller reads it (timed apart); best of %d" % reps}


def mid_size_line(ctx, M=20, N=165, R=9, T=2500):
    """The reference's real workload (experiments/full_experiment_script.py:57,87-113; pre_fitting_cmip6models.py:76-77): a
    collection of ~20 members, historical annual series N = 165, GPDTW1D fitted with n_optim_nits = 2000-2500.  One
    bayens_gp_fit_batch call on resident buffers: the fused mid-size kernel (csrc/medium.hip) by default, the general path
    (batched launches) beside it."""
    from bayesian_ensembling_amd.synthetic import synthetic_ensemble
    data, _ = synthetic_ensemble(M, N, R=R, R_obs=3, seed=0x165)
    X = np.ascontiguousarray(np.swapaxes(data, 1, 2))
    dX, dy, dnu = ctx.to_device(X), ctx.to_device(data.mean(axis=1)), ctx.to_device(data.var(axis=1))
    dmu, dcov, dchol = ctx.empty((M, N)), ctx.empty((M, N, N)), ctx.empty((M, N, N))
    dld, dhyp, dbasis = ctx.empty((M,)), ctx.empty((M, 2)), ctx.empty((M, 2, N))
    fit = lambda t: ctx.gp_fit_batch_dev(M, N, R, dX, dy, dnu, t, dmu, dcov, dchol, dld, dhyp, basis=dbasis)
    out = {}
    try:
        for key, min_batch in (("fused_kernel", 0), ("general_path", 10 ** 6)):
            ctx.set_medium_min_batch(min_batch)
            fit(2)
            ctx.timer_start(); fit(T); ms = ctx.timer_stop()
            tfl = M * fit_flops(N, R, T) / (ms * 1e-3) / 1e12
            out[key] = {"collection_fit_s": round(ms * 1e-3, 4), "us_per_iteration": round(ms * 1e3 / T, 1),
                        "fit_algorithmic_tflops": round(tfl, 3)}
            if (M, N, R, T) == (20, 165, 9, 2500):          # the timed result itself against the oracle's stored answer
                dev = check_known_answer("n165_T2500", dhyp.download()[0], dmu.download()[0], np.trace(dcov.download_rows(0, 1)[0]))
                out[key]["member0_vs_oracle_rel"] = float(f"{dev:.2e}")
    finally:
        ctx.set_medium_min_batch(0)
        for b in (dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis):
            b.free()
    cus = min(M, ctx.device_info()["compute_units"])
    per_cu_peak = F64_MFMA_PEAK_TFLOPS / ctx.device_info()["compute_units"]
    out["fused_kernel"]["frac_of_mfma_peak_of_the_cus_in_use"] = round(out["fused_kernel"]["fit_algorithmic_tflops"] / (per_cu_peak * cus), 4)
    out["workload"] = (f"{M} members x N={N} (historical annual series), R={R}, matern32, n_optim_nits={T}: the reference's "
                       "paper workload; fused_kernel = csrc/medium.hip, one workgroup (= one CU) per member; general_path = the "
                       "batched launches of the N=2000 machinery (round 2's route for this size)")
    out["speedup_vs_general_path"] = round(out["general_path"]["collection_fit_s"] / out["fused_kernel"]["collection_fit_s"], 3)
    return out


def paper_script_line(ctx, n_collections=14, members=20, T=2000, reps=2):
    """The fits of the reference's experiment script as ONE call: experiments/full_experiment_script.py:87-113 fits fourteen
    ModelCollections back to back (per scenario: the historical period, annual series N = 165, and the projection, N = 86), ~20
    members each with R = 3 ... 50 realisations, GPDTW1D with n_optim_nits = 2000 - on the CPU, member after member.  Here
    es.fit_collections stacks every member of equal length into one ragged launch (bayens_gp_fit_batch_ragged): 140 workgroups of
    the fused mid-size kernel and 140 of the fused small-member kernel instead of 14 x (distinct R) launches of <= 20.  Through the
    Python classes, host realisations in, posteriors resident; `one_collection_at_a_time` = `for mc: mc.fit(...)` beside it."""
    import warnings
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    from bayesian_ensembling_amd.synthetic import SplitMix64, synthetic_ensemble
    rng = SplitMix64(0x5C817)

    def collections():
        out = []
        for c in range(n_collections):
            N = 165 if c % 2 == 0 else 86
            tcoord = np.arange(N)
            pms = []
            for m in range(members):
                R = 3 + int(rng.uniform() * 48)                                     # 3 ... 50 (experiments/data: 3 ... 50)
                data, _ = synthetic_ensemble(1, N, R=R, seed=0x5C817 + 977 * c + m)
                pms.append(es.ProcessModel(DataArray(data[0], ("realisation", "time"), {"realisation": np.arange(R), "time": tcoord}),
                                           f"c{c}m{m}"))
            out.append(es.ModelCollection(pms))
        return out
    mcs = collections()
    Rs = [pm.n_realisations for mc in mcs for pm in mc.models]
    times = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for key in ("one_call", "one_collection_at_a_time"):
            best = None
            for _ in range(reps + 1 if key == "one_call" else 1):
                t0 = time.perf_counter()
                if key == "one_call":
                    es.fit_collections(mcs, es.GPDTW1D(), n_optim_nits=T, progress_bar=False)
                else:
                    for mc in mcs:
                        mc.fit(model=es.GPDTW1D(), n_optim_nits=T, progress_bar=False)
                ctx.sync()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            times[key] = best
            if key == "one_call":       # what the bit-equality tests say in -m gpu, asserted on the timed run too: finite, and the hypers in range
                hyp = np.array([[pm.distribution.hyperparameters["variance"], pm.distribution.hyperparameters["lengthscale"]]
                                for mc in mcs for pm in mc.models])
                mu0 = np.asarray(mcs[0][0].distribution.mean.values)
                assert np.isfinite(hyp).all() and (hyp > 0).all() and np.isfinite(mu0).all()
                ref_hyp = hyp.copy()
            else:
                hyp = np.array([[pm.distribution.hyperparameters["variance"], pm.distribution.hyperparameters["lengthscale"]]
                                for mc in mcs for pm in mc.models])
                assert np.array_equal(hyp, ref_hyp), "one call and collection-by-collection disagree"
    n = n_collections * members
    return {"wall_s": round(times["one_call"], 3), "member_fits_per_s": round(n / times["one_call"], 1),
            "one_collection_at_a_time_s": round(times["one_collection_at_a_time"], 3),
            "speedup_vs_one_collection_at_a_time": round(times["one_collection_at_a_time"] / times["one_call"], 2),
            "members": n, "realisations_min_max": [int(min(Rs)), int(max(Rs))], "checked": "hyper-parameters finite and identical "
            "between the two ways of calling (bit for bit); parity of the ragged launch itself: tests/test_gpu_ragged.py",
            "workload": f"{n_collections} collections x {members} members ({(n_collections + 1) // 2} of N=165, {n_collections // 2} of N=86), "
                        f"R drawn from 3..50 per member, matern32, n_optim_nits={T}: es.fit_collections(collections, GPDTW1D()) through "
                        f"the Python classes (best of {reps}), host realisations in, posteriors resident in HBM"}


def spawn_ranks(args):
    """bench.py --gpus N with no launcher: start N ranks here, one device each, before anything touches a GPU."""
    rdzv = tempfile.mkdtemp(prefix="bayens_rdzv_")
    nonce = os.urandom(8).hex()          # what the ranks of THIS run share: a rendezvous session of any other run is not theirs
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), BAYENS_RDZV=rdzv, BAYENS_RDZV_NONCE=nonce,
                   MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:           # one rank failed: the others would wait for it at the next collective
                    rc = code
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for q in procs:
            q.kill()
    raise SystemExit(rc)


def run_config5(args, ctx, rank, world, exchange, rdzv):
    """--config c5: grid points shard over the ranks with no data-path collective; each rank runs its own
    32 x grid-points field.  value = all ranks' fits / max-over-ranks time."""
    M, G, N, R, T = args.members, args.grid_points, 86, args.realisations, args.nits
    fw = FieldWorkload(ctx, M, G, N, R, T, args.export, seed=0xC5 + rank, f32=(args.dtype == "f32"))
    for _ in range(args.warmup):
        fw.step()
    ctx.sync()
    if exchange is not None:
        exchange.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fw.step()
    ctx.sync()
    if exchange is not None:
        exchange.barrier()
    elapsed = time.perf_counter() - t0
    if exchange is not None:
        elapsed = exchange.max_float(elapsed)
        exchange.close()
        rdzv.close()
    if rank != 0:
        return
    d = fw.measure(1, 0)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline_small(N, R, T, 400)
    out = {"metric": METRIC, "value": round(fw.B * world * args.steps / elapsed, 1), "unit": "member-fits/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"BASELINE config 5: {M} members x {G} grid points per GPU of independent GP fits (N={N} annual "
                                  f"steps, R={R}, matern32, n_optim_nits={T}, {args.export} export) + per-gridpoint "
                                  "LogLikelihoodWeight(c=1e-3) + Barycentre",
                      "preset": "c5", "members": M, "grid_points_per_gpu": G, "fits_per_gpu": fw.B, "timesteps": N,
                      "realisations": R, "n_optim_nits": T, "export": args.export,
                      "parallelism": f"grid points sharded x{world}, no collective" if world > 1 else "single GPU"},
           "roofline": d["roofline"], "roofline_hbm": d["roofline_hbm"], "cpu_baseline": cpu,
           "fit_kernel_ms": d["fit_kernel_ms"], "weighted_posterior_latency_ms": d["weighted_posterior_latency_ms"]}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json preset: c2 = 32 members/GPU fp64 matern32 (headline); c3 = fp32 matern52; "
                         "c4 = 16 members/GPU (128 members sharded 8-way); c5 = 32 members x 1728 grid points/GPU of "
                         "independent N=86 fits (96x144 points sharded 8-way, no collective)")
    ap.add_argument("--grid-points", type=int, default=1728, help="config 5: grid points per GPU")
    ap.add_argument("--export", default="reduced", choices=["reduced", "full"], help="config 5: what a fit exports")
    ap.add_argument("--members", type=int, default=None, help="members per GPU (default: the preset's)")
    ap.add_argument("--timesteps", type=int, default=2000)
    ap.add_argument("--realisations", type=int, default=8)
    ap.add_argument("--nits", type=int, default=2, help="n_optim_nits (2 = the reference's test scale)")
    ap.add_argument("--kernel", default=None, choices=["matern32", "matern52", "rbf"])
    ap.add_argument("--groups", type=int, default=0, help="member groups = HIP streams per fit batch (0: the library's default, by batch size)")
    ap.add_argument("--dtype", default=None, choices=["f64", "f32"],
                    help="f64 = the reference precision (the headline); f32 = BASELINE config 3 (1e-3 parity bar)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "file"],
                    help="N > 1 transport: rccl (the product) or file = host-staged rehearsal for several ranks on one GPU")
    ap.add_argument("--allow-host-exchange", action="store_true",
                    help="with --exchange rccl: if RCCL cannot be loaded or the communicator does not come up, fall back to "
                         "the host-staged exchange instead of exiting non-zero (rehearsals only; config.exchange says 'file')")
    ap.add_argument("--comm-init-timeout", type=float, default=180.0,
                    help="seconds a rank may spend inside ncclCommInitRank before it gives up with exit code 75")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra lines (T=500, config 3, RBF)")
    ap.add_argument("--cpu-members", type=int, default=2)
    args = ap.parse_args()
    preset = {"c2": (32, "matern32", "f64"), "c3": (32, "matern52", "f32"), "c4": (16, "matern32", "f64"),
              "c5": (32, "matern32", "f64")}[args.config]
    args.members = args.members or preset[0]
    args.kernel = args.kernel or preset[1]
    args.dtype = args.dtype or preset[2]

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                                       # never returns
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must match")

    from bayesian_ensembling_amd import _lib, parallel
    n_dev = _lib.device_count()
    if n_dev <= 0:
        raise SystemExit("no HIP device visible: the hot path needs an MI355X (there is no CPU fallback)")
    if local_rank >= n_dev and args.exchange == "rccl" and n_dev != 1:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {n_dev} device(s) visible")
    ctx = _lib.Context(local_rank % n_dev)  # (a launcher may expose a single device to every rank)
    ctx.set_groups(args.groups)
    exchange, rdzv = None, None
    force_comm = bool(os.environ.get("BAYENS_FORCE_COMM"))    # exercise the RCCL path with a single rank
    if world > 1 or force_comm:
        rdzv = parallel.FileRendezvous(parallel.FileRendezvous.directory_from_env(), rank, world)
        try:
            exchange, args.exchange, args.exchange_note = parallel.make_exchange(
                ctx, rdzv, args.exchange, allow_host_exchange=args.allow_host_exchange, init_timeout_s=args.comm_init_timeout)
        except (parallel.ExchangeUnavailable, TimeoutError) as e:
            # strict by default: a multi-rank line measured through /tmp files must never pass for an RCCL measurement
            # (TimeoutError: a peer never reached a rendezvous round at all - the same outcome, the same exit code)
            print(f"rank {rank}: {e}", file=sys.stderr, flush=True)
            raise SystemExit(3)
        if args.exchange == "rccl":
            r_, n_, _ = ctx.comm_info()
            if n_ != world or r_ != rank:
                raise SystemExit(f"rank {rank}: communicator reports rank {r_} of {n_}, expected {rank} of {world}")

    if args.config == "c5":
        run_config5(args, ctx, rank, world, exchange, rdzv)
        return
    M, N, R, T = args.members, args.timesteps, args.realisations, args.nits
    f32 = args.dtype == "f32"
    weighting = "w2_temporal" if args.config == "c3" else "loglik"      # C3 weights with ModelSimilarityWeight
    wl = Workload(ctx, M, N, R, T, args.kernel, f32, rank, world, exchange, args.groups, weighting=weighting)
    elapsed = wl.timed(args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    value = wl.Mtot * args.steps / elapsed
    rccl_lib = ctx.comm_info()[2] if exchange is not None and args.exchange == "rccl" else None
    rccl_ranks = ctx.comm_info()[1] if exchange is not None and args.exchange == "rccl" else 0
    if exchange is not None:        # every rank leaves the communicator here; what follows on rank 0 is local
        exchange.close()
        rdzv.close()
    if rank != 0:
        return

    # ---- untimed diagnostics on rank 0 (no collectives): latencies with HIP events, GEMM roofline, CPU baseline ----
    peak = F32_MFMA_PEAK_TFLOPS if f32 else F64_MFMA_PEAK_TFLOPS
    ctx.timer_start(); wl.fit(); fit_ms = ctx.timer_stop()
    wp_ms = None
    for _ in range(5):          # (a latency of tens of microseconds: one sample is at the mercy of the host)
        ctx.timer_start(); bad = wl.weighted_posterior(local_only=True); t_wp = ctx.timer_stop()
        wp_ms = t_wp if wp_ms is None else min(wp_ms, t_wp)
    ll_chol_ms = None
    if not f32:     # the stand-alone entry point (mu, chol, logdet -> rows) for comparison
        ctx.loglik_rows_dev(M, N, wl.R_obs, wl.dmu, wl.dchol, wl.dld, wl.dobs, wl.dll_loc)      # (first call grows its scratch)
        ctx.timer_start(); ctx.loglik_rows_dev(M, N, wl.R_obs, wl.dmu, wl.dchol, wl.dld, wl.dobs, wl.dll_loc); ll_chol_ms = ctx.timer_stop()
    roof = wl.roofline()
    ctx.set_groups(args.groups)
    fit_tfl = M * fit_flops(N, R, T, f32) / (fit_ms * 1e-3) / 1e12
    extra = {"fit_batch_ms": round(fit_ms, 3), "weighted_posterior_latency_ms": round(wp_ms, 4),
             "fit_algorithmic_tflops": round(fit_tfl, 3), "fit_frac_of_mfma_peak": round(fit_tfl / peak, 4),
             "fit_algorithmic_flop_per_member": fit_flops(N, R, T, f32),
             "fit_flop_note": "flop the device path executes by construction (fit_flops; DESIGN section 4): 13/3 N^3 per general "
                              "iteration, 5/3 N^3 in a first iteration that another follows (symmetric form), 10/3 N^3 for the "
                              "prediction + chol(cov); round 2 counted 9.6e10 for this workload, round 3 executes 7.5e10",
             "barycentre_not_converged": int(bad),
             "loglik_rows_from_chol_ms": None if ll_chol_ms is None else round(ll_chol_ms, 4),
             # > 0: the timed bayens_loglik_rows call(s) took the blocked-solve fallback of the chained launch, and the figure
             # above is the fallback's time (include/bayens.h, bayens_chain_stalls)
             "loglik_rows_chain_stalls": ctx.chain_stalls()}
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(N, R, T, args.cpu_members)

    # ---- extra lines (single GPU, the headline preset): API-default iteration count, config 3, the RBF wording ----
    if world == 1 and not args.no_extras and args.config == "c2" and (M, N, T, args.kernel, f32) == (32, 2000, 2, "matern32", False):
        ctx.timer_start(); info500 = wl.fit(T=500); t500_ms = ctx.timer_stop()
        assert not np.asarray(info500).any()
        dev500 = check_known_answer("c2_T500", wl.dhyp.download()[0], wl.dmu.download()[0], np.trace(wl.dcov.download_rows(0, 1)[0]))
        extra["n_optim_nits_500"] = {
            "member0_vs_oracle_rel": float(f"{dev500:.2e}"),
            "member_fits_per_s": round(M / (t500_ms * 1e-3), 3), "fit_batch_s": round(t500_ms * 1e-3, 3),
            "fit_algorithmic_tflops": round(M * fit_flops(N, R, 500) / (t500_ms * 1e-3) / 1e12, 3),
            "note": "the API default n_optim_nits=500 (ensembles/models.py:167), same 32 x 2000 batch, one fit batch"}
        wl.free()
        extra.update(api_path(ctx, M, N, R, T))
        extra["api_path_vs_resident_step"] = round(extra["api_path_ms_per_step"] / ms_per_step, 3)
        fw = FieldWorkload(ctx, 32, 1728, 86, 8, T, "reduced")
        c5 = fw.measure(3, 1)
        fw.free()
        fw = FieldWorkload(ctx, 32, 432, 86, 8, T, "full")
        c5f = fw.measure(3, 1)
        fw.free()
        c5["full_export"] = {k: c5f[k] for k in ("value", "fits_per_step", "fit_kernel_ms", "roofline", "roofline_hbm")}
        fw = FieldWorkload(ctx, 32, 432, 86, 8, T, "full", f32=True)
        c5s = fw.measure(3, 1)
        fw.free()
        c5["fp32_full_export"] = {k: c5s[k] for k in ("value", "fits_per_step", "fit_kernel_ms", "roofline_mfma")}
        c5["workload"] = ("BASELINE config 5, one GPU's share: 32 members x 1728 grid points (96x144 over 8 GPUs) of "
                          "independent N=86, R=8 fits + per-gridpoint LogLikelihoodWeight + Barycentre; full_export: 432 points")
        extra["config5"] = c5
        extra["n165_T2500"] = mid_size_line(ctx)
        extra["paper_script"] = paper_script_line(ctx)
        extra.update(api_path_c5(ctx, 32, 36, 48, 86, 8, T))
        extra["api_path_c5_vs_resident_step"] = round(extra["api_path_c5_ms_per_step"] / c5["ms_per_step"], 3)
        for key, kern, is32 in (("rbf_f64", "rbf", False), ("config3_f32_matern52", "matern52", True)):
            w2 = Workload(ctx, M, N, R, T, kern, is32, 0, 1, None, weighting="w2_temporal" if is32 else "loglik")
            el = w2.timed(3, 1)
            r2 = w2.roofline()
            extra[key] = {"value": round(M * 3 / el, 3), "unit": "member-fits/s", "ms_per_step": round(el / 3 * 1e3, 3),
                          "dtype": "f32" if is32 else "f64", "kernel": kern,
                          "weighting": "ModelSimilarityWeight(mode='temporal')" if is32 else "LogLikelihoodWeight",
                          "roofline": {k: r2[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                          "frac_on_padded_dims", "avg_launch_us")}}
            if is32:
                # BASELINE config 3's weighting step: ModelSimilarityWeight(mode='single') = M(M-1)/2 Gaussian W2 distances
                # between the full-covariance posteriors (weights.py:240-265); the fp32 factors are widened once, the polar
                # iterations run in fp64 (their first steps factor I + c X^T X with c ~ 1e9: not an fp32 computation)
                chol64, dws, dD = ctx.empty((M, N, N)), ctx.empty((M,)), ctx.empty((M, M))
                ctx.widen_f32_dev(M * N * N, w2.dchol, chol64)
                w2.local_rows()
                ctx.w2_single_dev(M, N, w2.dmu64, chol64, dws, dD)            # warm-up (workspace growth)
                ctx.sync()
                t0 = time.perf_counter()
                ctx.w2_single_dev(M, N, w2.dmu64, chol64, dws, dD)
                ctx.sync()
                w2_ms = (time.perf_counter() - t0) * 1e3
                pairs = M * (M - 1) // 2
                extra[key]["step_ms_with_w2_single"] = round(el / 3 * 1e3 + w2_ms, 1)    # fit + rows + barycentre + the W2 table
                extra[key]["w2_single"] = {"ms": round(w2_ms, 1), "pairs": pairs, "ms_per_pair": round(w2_ms / pairs, 3),
                                           "note": "ModelSimilarityWeight(mode='single') on the resident posteriors: "
                                                   "QDWH polar iteration per pair, fp64, all pairs in full batches of M"}
                for b in (chol64, dws, dD):
                    b.free()
            w2.free()

    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "member-fits/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{M}-member x {N}-timestep synthetic ensemble per GPU, R={R} realisations, "
                               f"{args.kernel} VGP fit (n_optim_nits={T}) + " +
                               ("ModelSimilarityWeight(mode='temporal')" if weighting == "w2_temporal" else "LogLikelihoodWeight(c=1e-3)") +
                               " + Barycentre",
                   "preset": args.config, "members_per_gpu": M, "members_total": M * world, "timesteps": N,
                   "realisations": R, "n_optim_nits": T, "kernel": args.kernel, "obs_realisations": wl.R_obs,
                   "member_groups": args.groups if args.groups else 1,
                   "parallelism": (f"member-sharded x{world}, one RCCL all-gather of packed [M_loc,3,N] rows per step"
                                   if world > 1 else "single GPU"),
                   "exchange": (args.exchange if world > 1 else None), "rccl_library": rccl_lib, "rccl_ranks": rccl_ranks,
                   "exchange_note": getattr(args, "exchange_note", None)},
        "roofline": roof, "cpu_baseline": cpu,
    }
    out.update(extra)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
