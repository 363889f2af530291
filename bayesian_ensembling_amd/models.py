"""Per-member posterior emulators.  `GPDTW1D` keeps the reference signature
(ensembles/models.py:160-230): fit(model, n_optim_nits=500, compile_objective=False, progress_bar=True)
-> Distribution, and raises NotImplementedError for more than (realisation, time) dimensions.

What runs where: the stage after the DTW barycentre - Matern-3/2 VGP with the fixed heteroskedastic Gaussian
likelihood, n_optim_nits x [natural-gradient(0.5) step, Adam(0.01) step], full-covariance prediction,
+diag(y_var), Cholesky/logdet of the result - is ONE batched libbayens call for all members that share a
realisation count (bayens_gp_fit_batch).  The DTW-barycentre target (tslearn in the reference, models.py:176-178)
is outside the accelerated boundary: `target` selects it.
"""
from __future__ import annotations

import numpy as np

from . import _lib
from . import distributions as dx
from .data import Distribution
from .labelled import ones_like
from .resident import ResidentBatch, ResidentField


def _blank_array(process_model):
    # xr.ones_like(model.model_data[0].drop("realisation")) * np.nan          (models.py:221-222)
    blank = ones_like(process_model.model_data[0].drop_vars("realisation")) * np.nan
    return blank.rename("blank")


class GPDTW1D:
    def __init__(self, name: str = "GPRegressor", target="mean", kernel: str = "matern32",
                 dtype: str = "float64", per_gridpoint: bool = False) -> None:
        """target: 'mean' (mean over realisations - the documented stand-in for the reference's tslearn DTW
        barycentre, which cannot be reproduced offline), 'dba' (DTW barycentre averaging with the semantics of the
        reference's own ensembles/dtwa.py performDBA, 10 iterations, on the GPU) or a callable (R,N) ndarray -> (N,).
        kernel: 'matern32' is the reference kernel (models.py:186); 'matern52' / 'rbf' are extensions.
        dtype: 'float64' is the reference precision (ensembles/__init__.py:10); 'float32' runs the fit in fp32
        (bayens_gp_fit_batch_f32, BASELINE config 3: 1e-3 relative) and hands the posterior back as float64."""
        self.name = name
        self.target = target
        self.kernel = kernel
        # per_gridpoint (opt-in, BASELINE config 5): (realisation, time, latitude, longitude) data is treated as one
        # independent (realisation, time) model per grid point - the reference's own 1-D fit at every point - instead of
        # raising NotImplementedError like the reference (models.py:171-174).  See fit_field.
        self.per_gridpoint = bool(per_gridpoint)
        if np.dtype(dtype) not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be 'float64' or 'float32'")
        self.dtype = np.dtype(dtype)

    # -- the boundary inputs (models.py:175-182) ------------------------------------------------------
    def _prologue(self, process_model):
        if process_model.model_data.ndim > 2:
            raise NotImplementedError("Not implemented for more than temporal dimensions. Use GPDTW3D instead")
        realisation_set = np.asarray(process_model.model_data.values, dtype=np.float64)
        if callable(self.target):
            y_mean = np.asarray(self.target(realisation_set), dtype=np.float64).reshape(-1)
        elif self.target == "mean":
            y_mean = realisation_set.mean(axis=0)
        elif self.target == "dba":
            y_mean = None                 # filled in batch by fit_collection (one device call per group)
        else:
            raise ValueError("target must be 'mean', 'dba' or a callable")
        y_var = np.var(realisation_set, axis=0)
        X = np.ascontiguousarray(realisation_set.T)
        return X, y_mean, y_var

    # -- per-gridpoint fields (BASELINE config 5) -------------------------------------------------------
    FIELD_CHUNK = 65536          # fits per launch of the fused small-member kernel
    FIELD_STAGES = 8             # upload / fit pipeline stages of a collection
    FIELD_WS_BYTES = 48 << 30    # general path (N > 254): workspace budget of one launch ...
    FIELD_WS_MATRICES = 8        # ... at this many padded N x N matrices per fit (csrc/fit_ws.h)

    def fit_field(self, process_models, n_optim_nits: int = 500):
        """Every (member, grid point) of a collection of (realisation, time, *space) models as an independent GP over
        time - models.py:175-220 applied to `model_data[:, :, lat, lon]` - in batched launches of the fused small-member
        kernel (N <= 94; 94 < N <= 254, the reference's historical horizon N = 165, through the fused mid-size kernel; longer
        series - the reference's monthly data, N = 1032 / 1980 - through the general path, a workspace-sized chunk of fits per
        launch) with the reduced export, resident in HBM as ONE ResidentField [M][G].  Members must share (R, T, space)."""
        ctx = _lib.default_context()
        shapes = {tuple(pm.model_data.shape) for pm in process_models}
        if len(shapes) != 1:
            raise ValueError("per_gridpoint: every member must have the same (realisation, time, *space) shape")
        if self.target != "mean" and not callable(self.target):
            raise NotImplementedError("per_gridpoint: target must be 'mean' or a callable")
        R, N = next(iter(shapes))[:2]
        G = int(np.prod(next(iter(shapes))[2:]))
        M = len(process_models)
        f32 = self.dtype != np.float64
        field = ResidentField(ctx, M, G, N)
        kid = _lib.KERNEL_IDS[self.kernel]
        B = M * G
        dX, dy, dnu = ctx.empty((B, N, R)), ctx.empty((B, N)), ctx.empty((B, N))
        draw = ctx.empty((M, R, N, G))
        # float32: the prologue and the field's tables stay float64 (they feed the float64 weight / scheme kernels); a chunk's
        # inputs are narrowed and its outputs widened on the device around the fp32 fit (bayens_gp_fit_batch_reduced_f32)
        # (general path: every fit of a chunk holds FIELD_WS_MATRICES padded N x N matrices of workspace)
        fused = N <= 254 and not (f32 and N > 94)
        Np = -(-N // 64) * 64
        per_launch = self.FIELD_CHUNK if fused else max(1, int(self.FIELD_WS_BYTES // (self.FIELD_WS_MATRICES * Np * Np * (4 if f32 else 8))))
        chunk = min(per_launch, B)
        sX, sy, snu, smu, svar, sbasis = ((ctx.empty((chunk, N, R), np.float32), ctx.empty((chunk, N), np.float32),
                                           ctx.empty((chunk, N), np.float32), ctx.empty((chunk, N), np.float32),
                                           ctx.empty((chunk, N), np.float32), ctx.empty((chunk, 2, N), np.float32))
                                          if f32 else (None,) * 6)
        worker = None
        try:
            # The realisations go up as they lie in memory ((realisation, time, *space), grid points contiguous); the
            # prologue of models.py:175-182 runs on the device for every grid point at once (bayens_field_prologue).
            # The collection is cut into stages of a few members: while one stage is being fitted, the next one's
            # realisations cross PCIe on a second stream (its own context, driven by a helper thread - ctypes releases
            # the GIL inside the blocking calls), so the upload (304 MB at config 5) hides behind the fits.
            row = R * N * G * 8
            # (stage sizes grow 1, 2, 4, ... up to M / FIELD_STAGES members: only the FIRST stage's upload is exposed, so it is
            #  the smallest; every later one hides behind the fits of the stage before it)
            per_stage = max(1, -(-M // self.FIELD_STAGES))
            stages, m0, size = [], 0, 1
            while m0 < M:
                m1 = min(M, m0 + min(size, per_stage))
                stages.append((m0, m1))
                m0, size = m1, size * 2
            up = _lib.upload_context(ctx)
            failure = []

            def upload(m0, m1):
                try:
                    for m in range(m0, m1):
                        data = np.ascontiguousarray(process_models[m].model_data.values, dtype=np.float64)
                        up._check(up.lib.bayens_memcpy_h2d(up.h, draw.offset(m * row), data.ctypes.data, data.nbytes))
                except Exception as e:       # noqa: BLE001 - re-raised on the main thread
                    failure.append(e)

            import threading
            worker = threading.Thread(target=upload, args=stages[0])
            worker.start()
            for k, (m0, m1) in enumerate(stages):
                worker.join()
                if failure:
                    raise failure[0]
                if k + 1 < len(stages):
                    worker = threading.Thread(target=upload, args=stages[k + 1])
                    worker.start()
                f0 = m0 * G
                ctx.field_prologue_dev(m1 - m0, R, N, G, draw.offset(m0 * row), dX.offset(f0 * N * R * 8),
                                       dy.offset(f0 * N * 8), dnu.offset(f0 * N * 8))
                if callable(self.target):   # a user-supplied barycentre replaces the mean target, grid point by grid point
                    y = np.empty(((m1 - m0) * G, N))
                    for m in range(m0, m1):
                        data = np.asarray(process_models[m].model_data.values, dtype=np.float64).reshape(R, N, G)
                        for g in range(G):
                            y[(m - m0) * G + g] = np.asarray(self.target(data[:, :, g]), dtype=np.float64).reshape(-1)
                    ctx._check(ctx.lib.bayens_memcpy_h2d(ctx.h, dy.offset(f0 * N * 8), y.ctypes.data, y.nbytes))
                for b0 in range(f0, m1 * G, chunk):
                    nb = min(m1 * G, b0 + chunk) - b0
                    if f32:
                        ctx.narrow_f64_dev(nb * N * R, dX.offset(b0 * N * R * 8), sX)
                        ctx.narrow_f64_dev(nb * N, dy.offset(b0 * N * 8), sy)
                        ctx.narrow_f64_dev(nb * N, dnu.offset(b0 * N * 8), snu)
                        ctx.gp_fit_batch_reduced_dev(nb, N, R, sX, sy, snu, n_optim_nits, smu, svar, field.logdet.offset(b0 * 8),
                                                     field.hyp.offset(b0 * 16), sbasis, kernel=kid, f32=True)
                        ctx.widen_f32_dev(nb * N, smu, field.mu.offset(b0 * N * 8))
                        ctx.widen_f32_dev(nb * N, svar, field.var.offset(b0 * N * 8))
                        ctx.widen_f32_dev(nb * 2 * N, sbasis, field.basis.offset(b0 * 2 * N * 8))
                        continue
                    ctx.gp_fit_batch_reduced_dev(nb, N, R, dX.offset(b0 * N * R * 8), dy.offset(b0 * N * 8),
                                                 dnu.offset(b0 * N * 8), n_optim_nits, field.mu.offset(b0 * N * 8),
                                                 field.var.offset(b0 * N * 8), field.logdet.offset(b0 * 8),
                                                 field.hyp.offset(b0 * 16), field.basis.offset(b0 * 2 * N * 8), kernel=kid)
            worker.join()
        finally:
            if worker is not None:
                worker.join()             # (never free a buffer an upload is still writing to)
            for d in (dX, dy, dnu, draw, sX, sy, snu, smu, svar, sbasis):
                if d is not None:
                    d.free()
        dists = []
        blank = _blank_array(process_models[0])         # (every member has the same shape and coordinates: one template)
        for m, pm in enumerate(process_models):
            d = Distribution(dim_array=blank, dist_type=dx.GridMultivariateNormalFullCovariance,
                             _resident=(field, m))
            d.losses = None
            dists.append(d)
        return dists

    def _prepare_group(self, ctx, pro, idxs, n_loss):
        """Resident inputs and outputs of ONE launch: the members `idxs` (equal length, any realisation counts)."""
        y = np.stack([pro[i][1] for i in idxs])
        nu = np.stack([pro[i][2] for i in idxs])
        B, N = y.shape
        # the posteriors stay in HBM (resident.py): inputs go up (a few MB), the fit runs on resident buffers, and
        # only the hyper-parameters and loss samples come back; mu / cov / chol are fetched if somebody reads them
        batch = ResidentBatch(ctx, B, N, self.dtype)
        if len({pro[i][0].shape[1] for i in idxs}) == 1:
            Rs, off, X = None, None, np.stack([pro[i][0] for i in idxs])
            R = X.shape[2]
        else:
            Rs, off, X = ctx.ragged_layout([pro[i][0] for i in idxs])
            R = None
        return {"idxs": idxs, "batch": batch, "B": B, "N": N, "R": R, "Rs": Rs, "off": off,
                "X": ctx.to_device(X, self.dtype), "y": ctx.to_device(y, self.dtype), "nu": ctx.to_device(nu, self.dtype),
                "loss": ctx.empty((B, n_loss)) if n_loss else None}

    @staticmethod
    def _launch_group(fctx, pr, n_optim_nits, kid, f32):
        """Queues (and waits for) the fit of a prepared group through the context `fctx` - the owner of the buffers, or a side
        context with its own stream when several groups run beside each other."""
        b = pr["batch"]
        if pr["Rs"] is None:
            fctx.gp_fit_batch_dev(pr["B"], pr["N"], pr["R"], pr["X"], pr["y"], pr["nu"], n_optim_nits, b.mu, b.cov, b.chol, b.logdet,
                                  b.hyp, kernel=kid, basis=b.basis, f32=f32, loss=pr["loss"])
        else:
            fctx.gp_fit_batch_ragged_dev(pr["B"], pr["N"], pr["Rs"], pr["off"], pr["X"], pr["y"], pr["nu"], n_optim_nits, b.mu, b.cov,
                                         b.chol, b.logdet, b.hyp, kernel=kid, basis=b.basis, loss=pr["loss"], f32=f32)

    def fit_collection(self, process_models, n_optim_nits: int = 500, compile_objective: bool = False,
                       progress_bar: bool = True):
        """All members at once.  The reference fits them one after the other (ensembles/data.py:385-395); their realisation
        counts differ (its pickled collections: R in {25, 9, 16}, {25, 25, 5, 6, 3, 5}; experiments/data: 3 ... 50).  Every
        member of equal length N goes into ONE launch whatever its R (bayens_gp_fit_batch_ragged{,_f32})."""
        del compile_objective   # accepted for signature compatibility (a tf.function knob)
        if self.per_gridpoint and all(pm.model_data.ndim > 2 for pm in process_models):
            return self.fit_field(process_models, n_optim_nits)
        # progress_bar: the reference shows the loss sampled every 25 iterations in a tqdm bar (models.py:211-215);
        # here the same samples are kept on each Distribution as `.losses` (and skipped when progress_bar=False)
        ctx = _lib.default_context()
        pro = [self._prologue(pm) for pm in process_models]
        if self.target == "dba":                      # the device DBA runs per distinct (N, R): its series are a dense [B][R][N]
            shapes = {}
            for i, (X, _, _) in enumerate(pro):
                shapes.setdefault(X.shape, []).append(i)
            for shape, idxs in shapes.items():
                y, _ = ctx.dba_batch(np.ascontiguousarray(np.stack([pro[i][0].T for i in idxs])), 10)
                for j, i in enumerate(idxs):
                    pro[i] = (pro[i][0], y[j], pro[i][2])
        f32 = self.dtype != np.float64
        groups = {}
        for i, (X, _, _) in enumerate(pro):
            groups.setdefault(X.shape[0], []).append(i)
        dists = [None] * len(process_models)
        kid = _lib.KERNEL_IDS[self.kernel]
        n_loss = (int(n_optim_nits) + 24) // 25 if progress_bar else 0
        # set-up of every group on the calling thread (allocation cache and uploads of `ctx`), then the launches
        prepared = [self._prepare_group(ctx, pro, idxs, n_loss) for idxs in groups.values()]
        try:
            if len(prepared) > 1:
                # Groups of different length are independent launches: a script's historical N = 165 and projection N = 86
                # collections (experiments/full_experiment_script.py:87-113) run BESIDE each other, each through its own context
                # = stream, driven by a helper thread (the blocking C call releases the GIL), so that the workgroups of one and of
                # the other share the 256 compute units instead of taking turns on them.
                import threading
                ctx.sync()                                   # the uploads ran on ctx's stream: every side stream must see them
                errors = []

                def launch(k):
                    try:
                        self._launch_group(_lib.side_context(ctx, k) if k else ctx, prepared[k], n_optim_nits, kid, f32)
                    except Exception as e:       # noqa: BLE001 - re-raised on the calling thread
                        errors.append(e)
                threads = [threading.Thread(target=launch, args=(k,)) for k in range(1, len(prepared))]
                for t in threads:
                    t.start()
                launch(0)
                for t in threads:
                    t.join()
                if errors:
                    raise errors[0]
            else:
                for pr in prepared:
                    self._launch_group(ctx, pr, n_optim_nits, kid, f32)
            for pr in prepared:
                batch, idxs = pr["batch"], pr["idxs"]
                batch.finish()
                hyp = batch.hyp.download()
                loss = pr["loss"].download() if pr["loss"] is not None else None
                for j, i in enumerate(idxs):
                    dists[i] = Distribution(dim_array=_blank_array(process_models[i]),
                                            dist_type=dx.MultivariateNormalFullCovariance, _resident=(batch, j))
                    dists[i].hyperparameters = {"variance": float(hyp[j, 0]), "lengthscale": float(hyp[j, 1])}
                    dists[i].losses = None if loss is None else loss[j].copy()
        finally:
            for pr in prepared:
                for d in (pr["X"], pr["y"], pr["nu"], pr["loss"]):
                    if d is not None:
                        d.free()
        return dists

    def fit(self, model, n_optim_nits: int = 500, compile_objective: bool = False, progress_bar: bool = True):
        return self.fit_collection([model], n_optim_nits, compile_objective, progress_bar)[0]


class MeanFieldApproximation:
    """ensembles/models.py:73-132.  The reference returns dx.Normal(mean, scale = variance) of the realisation set
    computed BEFORE its optimiser loop - the optimised parameters are discarded (models.py:129-131) -
    so the loop is not run here; `optimiser`, `n_optim_nits` and `compile_objective` are accepted and ignored."""

    def __init__(self, name="MeanFieldModel"):
        self.name = name

    def fit(self, model, optimiser=None, n_optim_nits: int = 500, compile_objective: bool = False) -> Distribution:
        del optimiser, n_optim_nits, compile_objective
        R = model.n_realisations
        data = np.asarray(model.model_data.values, dtype=np.float64).reshape(1, R, -1)
        mean, var = _lib.default_context().multi_model_mean(data, np.array([R], dtype=np.int32))
        blank = ones_like(model.model_data[0].drop_vars("realisation")) * np.nan
        return Distribution(mu=mean, covariance=var, dim_array=blank.rename("blank"), dist_type=dx.Normal)


def fit_collections(collections, model=None, **kwargs):
    """Fits SEVERAL ModelCollections at once: experiments/full_experiment_script.py:87-113 calls `ModelCollection.fit` on
    fourteen collections back to back (one per scenario and period; ~20 members each).  Members are independent
    (ensembles/data.py:391), so every member of equal length - whichever collection it belongs to, whatever its realisation
    count - is one workgroup of ONE launch: ~280 members fill the 256 compute units where 20 leave 236 of them idle.
    Same result as `for mc in collections: mc.fit(model, **kwargs)`; `model` must offer `fit_collection` (GPDTW1D)."""
    import warnings
    model = GPDTW1D() if model is None else model
    if not hasattr(model, "fit_collection"):
        for mc in collections:
            mc.fit(model, **kwargs)
        return
    members = [pm for mc in collections for pm in mc.models]
    if any(pm.distribution != None for pm in members):  # noqa: E711  (data.py:388-390)
        warnings.warn("Removing the model's previously learnt distribution")
    for pm, dist in zip(members, model.fit_collection(members, **kwargs)):
        pm.distribution = dist
