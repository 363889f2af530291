"""ctypes binding of libbayens.so (include/bayens.h).  The library is the product: there is no CPU
fallback - if it cannot be loaded, or no MI355X is visible, every hot-path call raises."""
from __future__ import annotations

import ctypes as C
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# BAYENS_LIB: a variant build (build.py, e.g. the bounds-checked mid-size kernel) instead of the product library - diagnostics only
LIB_PATH = os.environ.get("BAYENS_LIB") or os.path.join(HERE, "libbayens.so")
HEADER = os.path.join(HERE, "..", "include", "bayens.h")            # the drop-in boundary
DIAG_HEADER = os.path.join(HERE, "..", "include", "bayens_diag.h")  # diagnostics / benchmarks (not part of the boundary)

HOST, DEVICE = 0, 1
MATERN32, MATERN52, RBF = 0, 1, 2
KERNEL_IDS = {"matern32": MATERN32, "matern52": MATERN52, "rbf": RBF}

GEMM_TA, GEMM_TB, GEMM_TRI_OUT, GEMM_KTRIM_TN_LOWER, GEMM_KTRIM_A_LOWER, GEMM_ZERO_SKIPPED = 1, 2, 4, 8, 16, 32

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p


class BayensError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libbayens error {code}: {msg}")
        self.code = code


_lib = None


def declared_symbols():
    """Every function name include/bayens.h declares (used by the CPU test that checks the exports)."""
    txt = open(HEADER).read() + open(DIAG_HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(bayens_[a-z0-9_]+)\s*\(", txt)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BayensError(-3, f"{LIB_PATH} is missing: build it with `python -m bayesian_ensembling_amd.build` "
                              "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.bayens_version.restype = C.c_char_p
    lib.bayens_last_error.restype = C.c_char_p
    lib.bayens_last_error.argtypes = [_vp]
    lib.bayens_device_count.argtypes = [_ip]
    lib.bayens_ctx_create.argtypes = [C.c_int, C.POINTER(_vp)]
    lib.bayens_ctx_destroy.argtypes = [_vp]
    lib.bayens_device_info.argtypes = [_vp, C.c_char_p, C.c_int, _ip, C.POINTER(C.c_size_t)]
    lib.bayens_malloc.argtypes = [_vp, C.c_size_t, C.POINTER(_vp)]
    lib.bayens_free.argtypes = [_vp, _vp]
    lib.bayens_memcpy_h2d.argtypes = [_vp, _vp, _vp, C.c_size_t]
    lib.bayens_memcpy_d2h.argtypes = [_vp, _vp, _vp, C.c_size_t]
    lib.bayens_sync.argtypes = [_vp]
    lib.bayens_memcpy_d2d.argtypes = [_vp, _vp, _vp, C.c_size_t]
    lib.bayens_timer_start.argtypes = [_vp]
    lib.bayens_timer_stop.argtypes = [_vp, _dp]
    lib.bayens_profile_enable.argtypes = [_vp, C.c_int]
    lib.bayens_set_groups.argtypes = [_vp, C.c_int]
    lib.bayens_profile_stages.argtypes = [_vp, C.c_char_p, C.c_int]
    lib.bayens_profile_shapes.argtypes = [_vp, C.c_char_p, C.c_int]
    lib.bayens_profile_get.argtypes = [_vp, C.c_int, _dp, _dp, C.POINTER(C.c_long)]
    lib.bayens_profile_bytes.argtypes = [_vp, C.c_int, _dp]
    lib.bayens_gp_fit_batch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                        C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_gp_fit_batch_f32.argtypes = lib.bayens_gp_fit_batch.argtypes
    lib.bayens_gp_fit_batch_ragged.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int,
                                               C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_gp_fit_batch_ragged_f32.argtypes = lib.bayens_gp_fit_batch_ragged.argtypes
    lib.bayens_gp_fit_batch_reduced.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int,
                                                C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_transpose_rows.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.bayens_field_prologue.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_mvn_factor_batch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_loglik_rows_basis.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_loglik_rows_basis_grouped.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_extract_diag.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.bayens_weighted_posterior.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                              C.c_double, C.c_double, C.c_double, _vp, _vp, _vp, C.POINTER(C.c_int)]
    lib.bayens_extract_diag_f32.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp]
    lib.bayens_widen_f32.argtypes = [_vp, C.c_int, C.c_size_t, _vp, _vp]
    lib.bayens_narrow_f64.argtypes = [_vp, C.c_int, C.c_size_t, _vp, _vp]
    lib.bayens_gp_fit_batch_reduced_f32.argtypes = lib.bayens_gp_fit_batch_reduced.argtypes
    lib.bayens_loglik_rows.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_mvn_log_prob.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_normal_loglik_rows.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_weights_normalise.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_double, C.c_int, _vp, _vp]
    lib.bayens_w2_temporal.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.bayens_w2_spatial.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.bayens_w2_single.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_barycentre.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_double, C.c_double,
                                      _vp, _vp, _ip]
    lib.bayens_weighted_model_mean.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_score_weights.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_inverse_square_weights.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp]
    lib.bayens_multi_model_mean.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp]
    lib.bayens_dba_batch.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp]
    lib.bayens_comm_available.argtypes = [_vp]
    lib.bayens_comm_unique_id.argtypes = [_vp, _vp]
    lib.bayens_comm_init.argtypes = [_vp, C.c_int, C.c_int, _vp]
    lib.bayens_comm_destroy.argtypes = [_vp]
    lib.bayens_allgather_rows.argtypes = [_vp, _vp, _vp, C.c_int, C.c_int]
    lib.bayens_comm_info.argtypes = [_vp, _ip, _ip, C.c_char_p, C.c_int]
    lib.bayens_shard_bounds.argtypes = [C.c_int, C.c_int, C.c_int, _ip, _ip]
    lib.bayens_allgather_member_rows.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.bayens_comm_max.argtypes = [_vp, _dp]
    lib.bayens_w2_single_sharded.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]
    lib.bayens_test_gemm.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_int, _vp,
                                     C.c_int, C.c_double, _vp, C.c_int, C.c_int, C.c_double]
    lib.bayens_bench_gemm.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    lib.bayens_bench_leaf.argtypes = [_vp, C.c_int, C.c_int, _dp, _vp]
    lib.bayens_bench_potrf.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    lib.bayens_bench_trsm.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    lib.bayens_bench_mfma_peak.argtypes = [_vp, C.c_int, _dp]
    lib.bayens_bench_mfma_lds.argtypes = [_vp, C.c_int, _dp]
    lib.bayens_test_potrf.argtypes = [_vp, C.c_int, C.c_int, _vp, _vp]
    lib.bayens_small_stamps.argtypes = [_vp, C.c_int, _vp]
    lib.bayens_bench_diag16.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _dp]
    lib.bayens_test_trsm.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int]
    lib.bayens_set_medium_min_batch.argtypes = [_vp, C.c_int]
    lib.bayens_set_fit_route.argtypes = [_vp, C.c_int]
    lib.bayens_set_gemm_tile64_max.argtypes = [_vp, C.c_int]
    lib.bayens_chain_stalls.argtypes = [_vp, C.POINTER(C.c_long)]
    lib.bayens_small_wave_stamps.argtypes = [_vp, _vp]
    lib.bayens_potrf256_stamps.argtypes = [_vp, _vp]
    lib.bayens_test_member_rows_exchange.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]
    _lib = lib
    return lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, (DeviceBuffer, DeviceView)):
        return a.ptr
    if isinstance(a, int):
        return a
    return a.ctypes.data


class DeviceBuffer:
    """A block of HBM owned by a Context (bayens_malloc / bayens_free)."""

    def __init__(self, ctx, nbytes, shape=None, dtype=np.float64):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.shape = shape
        self.dtype = np.dtype(dtype)
        self.ptr = ctx._take_cached(self.nbytes)        # a block of exactly this size freed earlier, if there is one
        if self.ptr is None:
            p = _vp()
            rc = ctx.lib.bayens_malloc(ctx.h, self.nbytes, C.byref(p))
            if rc != 0 and ctx._cache:                  # HBM may be full of blocks the cache is holding on to: give them back, once
                ctx.trim_cache()
                rc = ctx.lib.bayens_malloc(ctx.h, self.nbytes, C.byref(p))
            ctx._check(rc)
            self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        assert arr.nbytes <= self.nbytes
        self.ctx._check(self.ctx.lib.bayens_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, shape=None):
        shape = shape if shape is not None else self.shape
        out = np.empty(shape, dtype=self.dtype)
        self.ctx._check(self.ctx.lib.bayens_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def offset(self, nbytes):
        return self.ptr + int(nbytes)

    def download_rows(self, first, count=1):
        """Rows [first, first + count) of the leading axis as a host array of shape (count,) + shape[1:]."""
        sub = tuple(self.shape[1:])
        out = np.empty((count,) + sub, dtype=self.dtype)
        row_bytes = int(np.prod(sub, dtype=np.int64)) * self.dtype.itemsize if sub else self.dtype.itemsize
        self.ctx._check(self.ctx.lib.bayens_memcpy_d2h(self.ctx.h, out.ctypes.data, self.ptr + first * row_bytes, out.nbytes))
        return out

    def copy_rows_from(self, dst_first, src, src_first, count=1):
        """Device-to-device copy of `count` leading-axis rows of `src` into this buffer (queued, no host wait)."""
        sub = tuple(self.shape[1:])
        row_bytes = int(np.prod(sub, dtype=np.int64)) * self.dtype.itemsize if sub else self.dtype.itemsize
        self.ctx._check(self.ctx.lib.bayens_memcpy_d2d(self.ctx.h, self.ptr + dst_first * row_bytes,
                                                       src.ptr + src_first * row_bytes, count * row_bytes))

    def free(self):
        if self.ptr:
            # (a buffer that outlives its context - Context.close() has destroyed the device context and everything it owned - has
            #  nothing left to free)
            if getattr(self.ctx, "h", None) is not None and not self.ctx._give_cached(self.nbytes, self.ptr):
                self.ctx.lib.bayens_free(self.ctx.h, self.ptr)
            self.ptr = None


class DeviceView:
    """A pointer into somebody else's DeviceBuffer (no ownership): what a device-resident argument of a Context call
    needs - `.ptr` and `.shape`."""

    def __init__(self, ptr, shape):
        self.ptr, self.shape = int(ptr), tuple(shape)


class Context:
    """One libbayens context = one HIP device + stream.  Raises BayensError when no GPU is usable."""

    def __init__(self, device=0):
        self.lib = load()
        h = _vp()
        rc = self.lib.bayens_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise BayensError(rc, "bayens_ctx_create failed (no usable HIP device: the hot path needs an MI355X; "
                                  "there is no CPU fallback)")
        self.h = h
        self.device = device
        # Freed blocks are kept for re-use by size.  A block may be handed out again the moment its owner lets go of it because
        # every operation on it is ordered on the context's one stream - with ONE exception the callers must close themselves:
        # a buffer written through upload_context()'s second stream (fit_field's `draw`) is only freed after the uploading thread
        # has been joined and the synchronous copies it made have returned.  hipMalloc / hipFree cost a millisecond or more each and
        # hipFree waits for the device: through the plug-in classes a weighting step allocated and freed half a dozen tables
        # around 0.2 ms of kernels.  Bounded: at most CACHE_BYTES in all, nothing above CACHE_MAX_BLOCK.
        self._cache = {}
        self._cache_bytes = 0

    CACHE_BYTES = 3 << 30
    CACHE_MAX_BLOCK = 1 << 30

    def _take_cached(self, nbytes):
        blocks = self._cache.get(nbytes)
        if blocks:
            self._cache_bytes -= nbytes
            return blocks.pop()
        return None

    def _give_cached(self, nbytes, ptr):
        if getattr(self, "h", None) is None or nbytes > self.CACHE_MAX_BLOCK:
            return False
        while self._cache_bytes + nbytes > self.CACHE_BYTES and self._cache:       # make room: largest size class first
            size = max(self._cache)
            blocks = self._cache[size]
            self.lib.bayens_free(self.h, blocks.pop())
            self._cache_bytes -= size
            if not blocks:
                del self._cache[size]
        if self._cache_bytes + nbytes > self.CACHE_BYTES:
            return False
        self._cache.setdefault(nbytes, []).append(ptr)
        self._cache_bytes += nbytes
        return True

    def trim_cache(self):
        """Returns every cached block to the device."""
        for size, blocks in self._cache.items():
            for ptr in blocks:
                self.lib.bayens_free(self.h, ptr)
        self._cache, self._cache_bytes = {}, 0

    def close(self):
        if getattr(self, "h", None):
            self.trim_cache()
            self.lib.bayens_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise BayensError(rc, self.lib.bayens_last_error(self.h).decode())

    # -- plumbing ---------------------------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        hbm = C.c_size_t()
        self._check(self.lib.bayens_device_info(self.h, name, 256, C.byref(cus), C.byref(hbm)))
        return {"name": name.value.decode(), "compute_units": cus.value, "hbm_bytes": hbm.value}

    def empty(self, shape, dtype=np.float64):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        return DeviceBuffer(self, max(n, 8), tuple(shape), dtype)

    def to_device(self, arr, dtype=np.float64):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        return self.empty(arr.shape, dtype).upload(arr)

    def sync(self):
        self._check(self.lib.bayens_sync(self.h))

    def timer_start(self):
        self._check(self.lib.bayens_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        self._check(self.lib.bayens_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def set_groups(self, n):
        """Member groups (HIP streams) a fit batch is split over; 1 serialises every kernel."""
        self._check(self.lib.bayens_set_groups(self.h, int(n)))

    def set_medium_min_batch(self, n):
        """Batch size from which members with 94 < N <= 254 take the fused mid-size kernel (1 = always, huge = never,
        0 = the default choice by size)."""
        self._check(self.lib.bayens_set_medium_min_batch(self.h, int(n)))

    def chain_stalls(self):
        """bayens_loglik_rows calls of this context that fell back from the chained substitution to the blocked solve."""
        n = C.c_long()
        self._check(self.lib.bayens_chain_stalls(self.h, C.byref(n)))
        return int(n.value)

    def set_gemm_tile64_max(self, max_workgroups):
        """-1: measured defaults; 0: the 128 x 128 kernel always; large: the 64 x 64-tile kernel always (fp64 context)."""
        self._check(self.lib.bayens_set_gemm_tile64_max(self.h, int(max_workgroups)))

    def set_fit_route(self, route):
        """1: every fp64 fit takes the general path whatever its size; 0: the fused kernels by size (default)."""
        self._check(self.lib.bayens_set_fit_route(self.h, int(route)))

    def profile_enable(self, on=True):
        self._check(self.lib.bayens_profile_enable(self.h, 1 if on else 0))

    def profile_stages(self):
        buf = C.create_string_buffer(8192)
        self._check(self.lib.bayens_profile_stages(self.h, buf, 8192))
        return buf.value.decode()

    def profile_shapes(self):
        buf = C.create_string_buffer(65536)
        self._check(self.lib.bayens_profile_shapes(self.h, buf, 65536))
        return buf.value.decode()

    def profile_get(self, family=0):
        ms, fl, n = C.c_double(), C.c_double(), C.c_long()
        self._check(self.lib.bayens_profile_get(self.h, family, C.byref(ms), C.byref(fl), C.byref(n)))
        by = C.c_double()
        self._check(self.lib.bayens_profile_bytes(self.h, family, C.byref(by)))
        return {"ms": ms.value, "flop": fl.value, "launches": n.value, "bytes": by.value}

    # -- hot path (host arrays in, host arrays out) --------------------------------------------------
    def gp_fit_batch(self, X, y_mean, y_var, n_iters, kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6,
                     want_chol=True, want_loss=False, dtype=np.float64):
        """dtype float64 (the reference precision) or float32 (bayens_gp_fit_batch_f32: fp32 storage and products;
        mu / cov / chol / basis come back as float32, logdet / hyp / loss stay float64)."""
        dt = np.dtype(dtype)
        if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        fn = self.lib.bayens_gp_fit_batch if dt == np.float64 else self.lib.bayens_gp_fit_batch_f32
        X = np.ascontiguousarray(X, dtype=dt)
        B, N, R = X.shape
        y_mean = np.ascontiguousarray(y_mean, dtype=dt).reshape(B, N)
        y_var = np.ascontiguousarray(y_var, dtype=dt).reshape(B, N)
        mu = np.empty((B, N), dtype=dt)
        cov = np.empty((B, N, N), dtype=dt)
        chol = np.empty((B, N, N), dtype=dt) if want_chol else None
        logdet = np.empty(B) if want_chol else None
        hyp = np.empty((B, 2))
        basis = np.empty((B, 2, N), dtype=dt) if want_chol else None
        info = np.zeros(B, dtype=np.int32)
        loss = np.empty((B, (int(n_iters) + 24) // 25)) if want_loss and n_iters > 0 else None
        rc = fn(self.h, HOST, B, N, R, _ptr(X), _ptr(y_mean), _ptr(y_var), int(kernel),
                                          int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(cov), _ptr(chol),
                                          _ptr(logdet), _ptr(hyp), _ptr(loss), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return {"mu": mu, "cov": cov, "chol": chol, "logdet": logdet, "hyp": hyp, "info": info, "basis": basis,
                "loss": loss}

    @staticmethod
    def ragged_layout(Xs):
        """Members with different realisation counts -> (R [B] int32, x_off [B] uint64, X: the members' (N, R_b) blocks one behind
        the other) - the layout bayens_gp_fit_batch_ragged takes."""
        Xs = [np.ascontiguousarray(x, dtype=np.float64) for x in Xs]
        N = Xs[0].shape[0]
        if any(x.ndim != 2 or x.shape[0] != N for x in Xs):
            raise ValueError("every member must be an (N, R_b) array with the same N")
        R = np.array([x.shape[1] for x in Xs], dtype=np.int32)
        off = np.zeros(len(Xs), dtype=np.uint64)
        off[1:] = np.cumsum(N * R.astype(np.uint64))[:-1]
        return R, off, np.concatenate([x.ravel() for x in Xs])

    def gp_fit_batch_ragged(self, Xs, y_mean, y_var, n_iters, kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6,
                            want_chol=True, want_loss=False):
        """Xs: a list of (N, R_b) arrays - members of equal N whose realisation counts differ - fitted in ONE launch
        (bayens_gp_fit_batch_ragged; ensembles/data.py:385-395 fits them one after the other)."""
        R, off, X = self.ragged_layout(Xs)
        B, N = len(Xs), Xs[0].shape[0]
        y_mean, y_var = _f64(y_mean).reshape(B, N), _f64(y_var).reshape(B, N)
        mu, cov = np.empty((B, N)), np.empty((B, N, N))
        chol = np.empty((B, N, N)) if want_chol else None
        logdet = np.empty(B) if want_chol else None
        hyp = np.empty((B, 2))
        basis = np.empty((B, 2, N)) if want_chol else None
        info = np.zeros(B, dtype=np.int32)
        loss = np.empty((B, (int(n_iters) + 24) // 25)) if want_loss and n_iters > 0 else None
        rc = self.lib.bayens_gp_fit_batch_ragged(self.h, HOST, B, N, _ptr(R), _ptr(off), _ptr(X), _ptr(y_mean), _ptr(y_var),
                                                 int(kernel), int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(cov), _ptr(chol),
                                                 _ptr(logdet), _ptr(hyp), _ptr(loss), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return {"mu": mu, "cov": cov, "chol": chol, "logdet": logdet, "hyp": hyp, "info": info, "basis": basis, "loss": loss}

    def gp_fit_batch_ragged_dev(self, B, N, R, x_off, X, y_mean, y_var, n_iters, mu, cov, chol=None, logdet=None, hyp=None,
                                kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6, basis=None, loss=None, f32=False):
        """Device-resident buffers (X: the concatenated blocks); R / x_off are HOST arrays (int32 / uint64, see ragged_layout).
        f32=True: X, y_mean, y_var, mu, cov, chol, basis are float32 buffers (bayens_gp_fit_batch_ragged_f32)."""
        R = np.ascontiguousarray(R, dtype=np.int32)
        x_off = np.ascontiguousarray(x_off, dtype=np.uint64)
        info = np.zeros(B, dtype=np.int32)
        fn = self.lib.bayens_gp_fit_batch_ragged_f32 if f32 else self.lib.bayens_gp_fit_batch_ragged
        rc = fn(self.h, DEVICE, B, N, _ptr(R), _ptr(x_off), _ptr(X), _ptr(y_mean), _ptr(y_var),
                                                 int(kernel), int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(cov), _ptr(chol),
                                                 _ptr(logdet), _ptr(hyp), _ptr(loss), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return info

    def gp_fit_batch_dev(self, B, N, R, X, y_mean, y_var, n_iters, mu, cov, chol=None, logdet=None, hyp=None,
                         kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6, basis=None, f32=False, loss=None):
        """Device-resident buffers; f32=True: X, y_mean, y_var, mu, cov, chol, basis are float32 buffers.
        loss: device buffer [B][ceil(n_iters / 25)] of float64 for the training-loss samples, or None."""
        info = np.zeros(B, dtype=np.int32)
        fn = self.lib.bayens_gp_fit_batch_f32 if f32 else self.lib.bayens_gp_fit_batch
        rc = fn(self.h, DEVICE, B, N, R, _ptr(X), _ptr(y_mean), _ptr(y_var), int(kernel),
                                          int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(cov), _ptr(chol),
                                          _ptr(logdet), _ptr(hyp), _ptr(loss), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return info

    def gp_fit_batch_reduced(self, X, y_mean, y_var, n_iters, kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6, dtype=np.float64):
        """Reduced export: mu, var = diag(cov), logdet, hyp, basis (no covariance leaves the device).  N <= 254: the fused kernels;
        above that the general path.  dtype float32: the fp32 instance (fused small-member kernel for N <= 94, the fp32 general
        path above that); mu / var / basis come back as float32."""
        dt = np.dtype(dtype)
        fn = self.lib.bayens_gp_fit_batch_reduced if dt == np.float64 else self.lib.bayens_gp_fit_batch_reduced_f32
        X = np.ascontiguousarray(X, dtype=dt)
        B, N, R = X.shape
        y_mean, y_var = np.ascontiguousarray(y_mean, dtype=dt).reshape(B, N), np.ascontiguousarray(y_var, dtype=dt).reshape(B, N)
        mu, var, logdet, hyp = np.empty((B, N), dtype=dt), np.empty((B, N), dtype=dt), np.empty(B), np.empty((B, 2))
        basis, info = np.empty((B, 2, N), dtype=dt), np.zeros(B, dtype=np.int32)
        rc = fn(self.h, HOST, B, N, R, _ptr(X), _ptr(y_mean), _ptr(y_var), int(kernel),
                                                  int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(var), _ptr(logdet),
                                                  _ptr(hyp), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return {"mu": mu, "var": var, "logdet": logdet, "hyp": hyp, "basis": basis, "info": info}

    def gp_fit_batch_reduced_dev(self, B, N, R, X, y_mean, y_var, n_iters, mu, var, logdet=None, hyp=None, basis=None,
                                 kernel=MATERN32, gamma=0.5, lr=0.01, jitter=1e-6, f32=False):
        """f32=True: X, y_mean, y_var, mu, var, basis are float32 buffers (logdet / hyp stay float64)."""
        info = np.zeros(B, dtype=np.int32)
        fn = self.lib.bayens_gp_fit_batch_reduced_f32 if f32 else self.lib.bayens_gp_fit_batch_reduced
        rc = fn(self.h, DEVICE, B, N, R, _ptr(X), _ptr(y_mean), _ptr(y_var),
                                                  int(kernel), int(n_iters), gamma, lr, jitter, _ptr(mu), _ptr(var),
                                                  _ptr(logdet), _ptr(hyp), _ptr(info), _ptr(basis))
        if rc != 0:
            err = BayensError(rc, self.lib.bayens_last_error(self.h).decode())
            err.info = info
            raise err
        return info

    def transpose_rows_dev(self, nb, A, B, src, dst):
        """[nb][A][B] -> [nb][B][A] on resident buffers (out of place)."""
        self._check(self.lib.bayens_transpose_rows(self.h, DEVICE, nb, A, B, _ptr(src), _ptr(dst)))

    def field_prologue_dev(self, M, R, N, G, raw, X, y_mean, y_var):
        """models.py:175-182 at every grid point, on resident buffers: raw [M][R][N][G] -> X [M*G][N][R], y, nu [M*G][N]."""
        self._check(self.lib.bayens_field_prologue(self.h, DEVICE, M, R, N, G, _ptr(raw), _ptr(X), _ptr(y_mean), _ptr(y_var)))

    def mvn_factor_batch(self, cov, mu=None):
        """chol, logdet (and, with mu, the log-likelihood basis) of caller-supplied covariances."""
        cov = _f64(cov)
        B, N, _ = cov.shape
        chol = np.empty((B, N, N))
        logdet = np.empty(B)
        info = np.zeros(B, dtype=np.int32)
        basis = None
        if mu is not None:
            mu = _f64(mu).reshape(B, N)
            basis = np.empty((B, 2, N))
        self._check(self.lib.bayens_mvn_factor_batch(self.h, HOST, B, N, _ptr(cov), _ptr(mu), _ptr(chol), _ptr(logdet),
                                                     _ptr(basis), _ptr(info)))
        return (chol, logdet, basis) if mu is not None else (chol, logdet)

    def loglik_rows_basis(self, basis, logdet, obs):
        basis, logdet, obs = _f64(basis), _f64(logdet), _f64(obs)
        B, _, N = basis.shape
        out = np.empty((B, N))
        self._check(self.lib.bayens_loglik_rows_basis(self.h, HOST, B, N, obs.shape[0], _ptr(basis), _ptr(logdet),
                                                      _ptr(obs), _ptr(out)))
        return out

    def loglik_rows_basis_dev(self, B, N, R_obs, basis, logdet, obs, ll_mean):
        self._check(self.lib.bayens_loglik_rows_basis(self.h, DEVICE, B, N, R_obs, _ptr(basis), _ptr(logdet),
                                                      _ptr(obs), _ptr(ll_mean)))

    def loglik_rows_basis_grouped(self, basis, logdet, obs, loc=HOST, out=None):
        """Field of posteriors [M][G]: basis [B,2,N], logdet [B], obs [G,R_obs,N] -> ll_mean [B,N] (B = M * G)."""
        if loc == HOST:
            basis, logdet, obs = _f64(basis), _f64(logdet), _f64(obs)
            out = np.empty((basis.shape[0], basis.shape[2]))
        B, N = basis.shape[0], basis.shape[-1]
        G, R_obs = obs.shape[0], obs.shape[1]
        self._check(self.lib.bayens_loglik_rows_basis_grouped(self.h, loc, B, N, R_obs, G, _ptr(basis), _ptr(logdet),
                                                              _ptr(obs), _ptr(out)))
        return out

    def extract_diag(self, cov):
        cov = _f64(cov)
        B, N, _ = cov.shape
        var = np.empty((B, N))
        self._check(self.lib.bayens_extract_diag(self.h, HOST, B, N, _ptr(cov), _ptr(var)))
        return var

    def loglik_rows(self, mu, chol, logdet, obs, loc=HOST, out=None):
        if loc == HOST:
            mu, chol, logdet, obs = _f64(mu), _f64(chol), _f64(logdet), _f64(obs)
            B, N = mu.shape
            R_obs = obs.shape[0]
            out = np.empty((B, N))
            self._check(self.lib.bayens_loglik_rows(self.h, HOST, B, N, R_obs, _ptr(mu), _ptr(chol), _ptr(logdet),
                                                    _ptr(obs), _ptr(out)))
            return out
        raise ValueError("use loglik_rows_dev for resident buffers")

    def loglik_rows_dev(self, B, N, R_obs, mu, chol, logdet, obs, ll_mean):
        self._check(self.lib.bayens_loglik_rows(self.h, DEVICE, B, N, R_obs, _ptr(mu), _ptr(chol), _ptr(logdet),
                                                _ptr(obs), _ptr(ll_mean)))

    def mvn_log_prob(self, mu, chol, logdet, x, loc=HOST):
        """log N(x_r; mu_b, L_b L_b^T): mu [B,N], chol [B,N,N], logdet [B]; x [n_x,N] (shared) or [B,n_x,N] -> [B,n_x].
        loc=DEVICE: mu / chol / logdet are device buffers (x and the result stay host arrays)."""
        x = _f64(x)
        shared = x.ndim == 2
        if loc == HOST:
            mu, chol, logdet = _f64(mu), _f64(chol), _f64(logdet)
            B, N = mu.shape
        else:
            B, N = mu.shape[0], mu.shape[-1]
        n_x = x.shape[-2]
        out = np.empty((B, n_x))
        if loc == HOST:
            self._check(self.lib.bayens_mvn_log_prob(self.h, HOST, B, N, n_x, 1 if shared else 0, _ptr(mu), _ptr(chol),
                                                     _ptr(logdet), _ptr(x), _ptr(out)))
            return out
        dx_, dout = self.to_device(x), self.empty((B, n_x))
        try:
            self._check(self.lib.bayens_mvn_log_prob(self.h, DEVICE, B, N, n_x, 1 if shared else 0, _ptr(mu), _ptr(chol),
                                                     _ptr(logdet), _ptr(dx_), _ptr(dout)))
            return dout.download()
        finally:
            dx_.free()
            dout.free()

    def normal_loglik_rows(self, mu, scale, obs):
        mu, scale, obs = _f64(mu), _f64(scale), _f64(obs)
        B, N = mu.shape
        out = np.empty((B, N))
        self._check(self.lib.bayens_normal_loglik_rows(self.h, HOST, B, N, obs.shape[0], _ptr(mu), _ptr(scale),
                                                       _ptr(obs), _ptr(out)))
        return out

    def weights_normalise(self, ll_mean, c_std=1.0, return_raw=False):
        ll_mean = _f64(ll_mean)
        M, P = ll_mean.shape
        w = np.empty((M, P))
        raw = np.empty((M, P)) if return_raw else None
        self._check(self.lib.bayens_weights_normalise(self.h, HOST, M, P, _ptr(ll_mean), float(c_std), 0, _ptr(w),
                                                      _ptr(raw)))
        return (w, raw) if return_raw else w

    def weights_normalise_dev(self, M, P, ll_mean, c_std, w, w_raw=None):
        self._check(self.lib.bayens_weights_normalise(self.h, DEVICE, M, P, _ptr(ll_mean), float(c_std), 0, _ptr(w),
                                                      _ptr(w_raw)))

    def w2_temporal(self, mean, var):
        mean, var = _f64(mean), _f64(var)
        M, P = mean.shape
        w = np.empty((M, P))
        self._check(self.lib.bayens_w2_temporal(self.h, HOST, M, P, _ptr(mean), _ptr(var), _ptr(w)))
        return w

    def w2_spatial(self, mean, scale):
        """ModelSimilarityWeight(mode='spatial'), dx.Normal branch: mean, scale [M][T][G] -> w [M][G]."""
        mean, scale = _f64(mean), _f64(scale)
        M, T, G = mean.shape
        w = np.empty((M, G))
        self._check(self.lib.bayens_w2_spatial(self.h, HOST, M, T, G, _ptr(mean), _ptr(scale), _ptr(w)))
        return w

    def w2_temporal_dev(self, M, P, mean, var, w):
        self._check(self.lib.bayens_w2_temporal(self.h, DEVICE, M, P, _ptr(mean), _ptr(var), _ptr(w)))

    def w2_single_dev(self, M, N, mu, chol, w, D):
        """Full-covariance posteriors resident in HBM: mu [M][N], chol [M][N][N] device buffers."""
        self._check(self.lib.bayens_w2_single(self.h, DEVICE, M, N, _ptr(mu), _ptr(chol), None, _ptr(w), _ptr(D)))

    def w2_single(self, mu, chol=None, sd=None):
        """ModelSimilarityWeight(mode='single'): (w [M], D [M][M]) from full-covariance factors or stddevs."""
        mu = _f64(mu)
        M, N = mu.shape
        chol = None if chol is None else _f64(chol)
        sd = None if sd is None else _f64(sd)
        w = np.empty(M)
        D = np.empty((M, M))
        self._check(self.lib.bayens_w2_single(self.h, HOST, M, N, _ptr(mu), _ptr(chol), _ptr(sd), _ptr(w), _ptr(D)))
        return w, D

    def barycentre(self, w, mean, var, tol=1e-6, init_var=1.0):
        w, mean, var = _f64(w), _f64(mean), _f64(var)
        M, P = mean.shape
        w = np.ascontiguousarray(w.reshape(M, -1))
        mu = np.empty(P)
        v = np.empty(P)
        bad = C.c_int(0)
        self._check(self.lib.bayens_barycentre(self.h, HOST, M, P, _ptr(w), _ptr(mean), _ptr(var), tol, init_var,
                                               _ptr(mu), _ptr(v), C.byref(bad)))
        return mu, v, bad.value

    def barycentre_dev(self, M, P, w, mean, var, mu_out, var_out, tol=1e-6, init_var=1.0):
        bad = C.c_int(0)
        self._check(self.lib.bayens_barycentre(self.h, DEVICE, M, P, _ptr(w), _ptr(mean), _ptr(var), tol, init_var,
                                               _ptr(mu_out), _ptr(var_out), C.byref(bad)))
        return bad.value

    def weighted_posterior_dev(self, M, N, R_obs, basis, logdet, obs, mu, cov, w, mu_out, var_out, c_std=1e-3,
                               tol=1e-6, init_var=1.0):
        """LogLikelihoodWeight + Barycentre on resident buffers, one host synchronisation; returns n_not_converged."""
        bad = C.c_int()
        self._check(self.lib.bayens_weighted_posterior(self.h, DEVICE, M, N, R_obs, _ptr(basis), _ptr(logdet), _ptr(obs),
                                                       _ptr(mu), _ptr(cov), c_std, tol, init_var, _ptr(w), _ptr(mu_out),
                                                       _ptr(var_out), C.byref(bad)))
        return bad.value

    def weighted_posterior(self, basis, logdet, obs, mu, cov, c_std=1e-3, tol=1e-6, init_var=1.0):
        basis, logdet, obs, mu, cov = _f64(basis), _f64(logdet), _f64(obs), _f64(mu), _f64(cov)
        M, N = mu.shape
        w, bm, bv, bad = np.empty((M, N)), np.empty(N), np.empty(N), C.c_int()
        self._check(self.lib.bayens_weighted_posterior(self.h, HOST, M, N, obs.shape[0], _ptr(basis), _ptr(logdet),
                                                       _ptr(obs), _ptr(mu), _ptr(cov), c_std, tol, init_var, _ptr(w),
                                                       _ptr(bm), _ptr(bv), C.byref(bad)))
        return w, bm, bv, bad.value

    def extract_diag_dev(self, B, N, cov, var, f32=False):
        fn = self.lib.bayens_extract_diag_f32 if f32 else self.lib.bayens_extract_diag
        self._check(fn(self.h, DEVICE, B, N, _ptr(cov), _ptr(var)))

    def narrow_f64_dev(self, n, src, dst):
        self._check(self.lib.bayens_narrow_f64(self.h, DEVICE, int(n), _ptr(src), _ptr(dst)))

    def widen_f32_dev(self, n, src, dst):
        self._check(self.lib.bayens_widen_f32(self.h, DEVICE, int(n), _ptr(src), _ptr(dst)))

    def weighted_model_mean(self, data, R, w):
        data, w = _f64(data), _f64(w)
        M, R_max, P = data.shape
        R = np.ascontiguousarray(R, dtype=np.int32)
        mu = np.empty(P)
        v = np.empty(P)
        self._check(self.lib.bayens_weighted_model_mean(self.h, HOST, M, R_max, _ptr(R), P, _ptr(data), _ptr(w),
                                                        _ptr(mu), _ptr(v)))
        return mu, v

    def score_weights(self, kind, mean, var, obs):
        """kind 'crps' | 'ksd' -> (w [M][P], score [M][P])"""
        mean, var, obs = _f64(mean), _f64(var), _f64(obs)
        M, P = mean.shape
        w, sc = np.empty((M, P)), np.empty((M, P))
        self._check(self.lib.bayens_score_weights(self.h, HOST, {"crps": 0, "ksd": 1}[kind], M, P, obs.shape[0],
                                                  _ptr(mean), _ptr(var), _ptr(obs), _ptr(w), _ptr(sc)))
        return w, sc

    def inverse_square_weights(self, data, R, obs):
        data, obs = _f64(data), _f64(obs)
        M, R_max, P = data.shape
        R = np.ascontiguousarray(R, dtype=np.int32)
        w = np.empty((M, P))
        self._check(self.lib.bayens_inverse_square_weights(self.h, HOST, M, R_max, _ptr(R), P, _ptr(data), obs.shape[0],
                                                           _ptr(obs), _ptr(w)))
        return w

    def multi_model_mean(self, data, R):
        data = _f64(data)
        M, R_max, P = data.shape
        R = np.ascontiguousarray(R, dtype=np.int32)
        mu, v = np.empty(P), np.empty(P)
        self._check(self.lib.bayens_multi_model_mean(self.h, HOST, M, R_max, _ptr(R), P, _ptr(data), _ptr(mu), _ptr(v)))
        return mu, v

    def dba_batch(self, series, n_iterations=10):
        """DTW barycentre of every member's realisations: series [B][R][N] -> (center [B][N], medoid [B])."""
        series = _f64(series)
        B, R, N = series.shape
        center = np.empty((B, N))
        medoid = np.zeros(B, dtype=np.int32)
        self._check(self.lib.bayens_dba_batch(self.h, HOST, B, R, N, _ptr(series), int(n_iterations), _ptr(center),
                                              _ptr(medoid)))
        return center, medoid

    # -- RCCL ------------------------------------------------------------------------------------------
    def comm_available(self):
        """Raises unless librccl can be loaded (load-only probe: no id drawn, no communicator)."""
        self._check(self.lib.bayens_comm_available(self.h))

    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        self._check(self.lib.bayens_comm_unique_id(self.h, buf))
        return buf.raw

    def comm_init(self, rank, nranks, uid):
        """Joins the communicator.  RCCL prints a version banner on the C-level stdout while initialising; it is sent
        to stderr so that a program's stdout stays what the program itself prints (bench.py: ONE JSON line)."""
        buf = C.create_string_buffer(bytes(uid), 128)
        sys.stdout.flush()
        saved = os.dup(1)
        try:
            os.dup2(2, 1)
            rc = self.lib.bayens_comm_init(self.h, rank, nranks, buf)
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        self._check(rc)

    def comm_destroy(self):
        self._check(self.lib.bayens_comm_destroy(self.h))

    def comm_info(self):
        """(rank, nranks, path of the RCCL library in use); nranks 0 = no communicator."""
        r, n = C.c_int(), C.c_int()
        buf = C.create_string_buffer(512)
        self._check(self.lib.bayens_comm_info(self.h, C.byref(r), C.byref(n), buf, 512))
        return r.value, n.value, buf.value.decode()

    def allgather_member_rows(self, M, P, M_loc, ll_loc, mean_loc, var_loc, ll_all, mean_all, var_all):
        """The exchange of the sharded weighted posterior on resident rows (uneven shards allowed)."""
        self._check(self.lib.bayens_allgather_member_rows(self.h, M, P, M_loc, _ptr(ll_loc), _ptr(mean_loc),
                                                          _ptr(var_loc), _ptr(ll_all), _ptr(mean_all), _ptr(var_all)))

    def comm_max(self, x):
        """max over the ranks of one double; also a barrier (every rank's stream has drained when it returns)."""
        v = C.c_double(float(x))
        self._check(self.lib.bayens_comm_max(self.h, C.byref(v)))
        return v.value

    def w2_single_sharded_dev(self, M, N, mu, chol, w, D):
        """All members' mu / chol resident on every rank; pairs dealt over the communicator's ranks."""
        self._check(self.lib.bayens_w2_single_sharded(self.h, DEVICE, M, N, _ptr(mu), _ptr(chol), _ptr(w), _ptr(D)))

    def allgather_rows(self, local, all_, M_loc, P):
        self._check(self.lib.bayens_allgather_rows(self.h, _ptr(local), _ptr(all_), M_loc, P))

    # -- diagnostics -----------------------------------------------------------------------------------
    def test_gemm(self, A, B, Cm, alpha=1.0, beta=0.0, flags=0, diag_add=0.0):
        A, B = _f64(A), _f64(B)
        Cm = np.array(Cm, dtype=np.float64, order="C", copy=True)
        nb, m, n = Cm.shape
        k = A.shape[1] if (flags & GEMM_TA) else A.shape[2]
        self._check(self.lib.bayens_test_gemm(self.h, nb, m, n, k, alpha, _ptr(A), A.shape[2], _ptr(B), B.shape[2],
                                              beta, _ptr(Cm), Cm.shape[2], flags, diag_add))
        return Cm

    def bench_gemm(self, nb, m, n, k, flags=0, reps=10):
        ms = C.c_double()
        self._check(self.lib.bayens_bench_gemm(self.h, nb, m, n, k, flags, reps, C.byref(ms)))
        return ms.value

    def bench_leaf(self, nb=32, reps=50):
        us = C.c_double()
        st = np.zeros(16, dtype=np.uint64)
        self._check(self.lib.bayens_bench_leaf(self.h, nb, reps, C.byref(us), st.ctypes.data))
        return us.value, st

    def bench_mfma_peak(self, waves_per_simd=2):
        tf = C.c_double()
        self._check(self.lib.bayens_bench_mfma_peak(self.h, waves_per_simd, C.byref(tf)))
        return tf.value

    def bench_mfma_lds(self, mode=0):
        tf = C.c_double()
        self._check(self.lib.bayens_bench_mfma_lds(self.h, mode, C.byref(tf)))
        return tf.value

    def bench_trsm(self, nb=32, m=2048, n=2048, mode=0, rhs_lower=False, reps=10):
        """Diagnostic: microseconds per blocked triangular solve X op(L) = B (bayens_diag.h)."""
        us = C.c_double(0.0)
        self._check(self.lib.bayens_bench_trsm(self.h, nb, m, n, mode, int(rhs_lower), reps, C.byref(us)))
        return us.value

    def bench_potrf(self, nb=32, n=2048, reps=10):
        us = C.c_double()
        self._check(self.lib.bayens_bench_potrf(self.h, nb, n, reps, C.byref(us)))
        return us.value

    def bench_diag16(self, variant=0, waves=1, reps=200):
        cyc = C.c_double()
        self._check(self.lib.bayens_bench_diag16(self.h, variant, waves, reps, C.byref(cyc)))
        return cyc.value

    def small_stamps(self, enable=True, read=False):
        """Phase stamps of the fused small-member kernel (diagnostic); returns 32 uint64 when read=True."""
        out = np.zeros(32, dtype=np.uint64) if read else None
        self._check(self.lib.bayens_small_stamps(self.h, 1 if enable else 0, _ptr(out)))
        return out

    def small_wave_stamps(self):
        """Per-wave stamps of the last recorded small fit: (8, 16) uint64 (zeros where a build places no stamp)."""
        out = np.zeros(128, dtype=np.uint64)
        self._check(self.lib.bayens_small_wave_stamps(self.h, _ptr(out)))
        return out.reshape(8, 16)

    def potrf256_stamps(self):
        out = np.zeros(128, dtype=np.uint64)
        self._check(self.lib.bayens_potrf256_stamps(self.h, _ptr(out)))
        return out.reshape(8, 16)

    def test_potrf(self, A):
        A = np.array(A, dtype=np.float64, order="C", copy=True)
        nb, n, _ = A.shape
        info = np.zeros(nb, dtype=np.int32)
        self._check(self.lib.bayens_test_potrf(self.h, nb, n, _ptr(A), _ptr(info)))
        return A, info

    def test_member_rows_exchange(self, nranks, ll, mean, var):
        """pack (per fictitious rank) -> gathered image -> unpack of bayens_allgather_member_rows, without RCCL."""
        ll, mean, var = _f64(ll), _f64(mean), _f64(var)
        M, P = ll.shape
        out = [np.empty((M, P)) for _ in range(3)]
        self._check(self.lib.bayens_test_member_rows_exchange(self.h, M, P, int(nranks), _ptr(ll), _ptr(mean), _ptr(var),
                                                              _ptr(out[0]), _ptr(out[1]), _ptr(out[2])))
        return out

    def test_trsm(self, L, Bm, mode):
        L = _f64(L)
        Bm = np.array(Bm, dtype=np.float64, order="C", copy=True)
        nb, m, n = Bm.shape
        self._check(self.lib.bayens_test_trsm(self.h, nb, m, n, _ptr(L), _ptr(Bm), mode))
        return Bm


def device_count():
    n = C.c_int(0)
    load().bayens_device_count(C.byref(n))
    return n.value


_default_ctx = None
_upload_ctxs = {}


def upload_context(ctx):
    """A second context (its own HIP stream) on the device of `ctx`, for host-to-device copies that overlap kernels
    running on `ctx` (contexts are independent and may be driven from different threads; device memory is shared)."""
    c = _upload_ctxs.get(ctx.device)
    if c is None:
        c = _upload_ctxs[ctx.device] = Context(ctx.device)
    return c


_side_ctxs = {}


def side_context(ctx, k):
    """The k-th extra context (k >= 1; its own HIP stream and workspaces) on the device of `ctx`: launches queued through it run
    beside those of `ctx` (GPDTW1D.fit_collection gives every group of equal-length members its own).  Buffers of `ctx` may be
    passed to it - device memory is shared; the caller orders the two streams (Context.sync) where one reads what the other wrote."""
    key = (ctx.device, int(k))
    c = _side_ctxs.get(key)
    if c is None:
        c = _side_ctxs[key] = Context(ctx.device)
    return c


def default_context():
    """Process-wide context on the device LOCAL_RANK selects (one process per GPU)."""
    global _default_ctx
    if _default_ctx is None:
        n = max(device_count(), 1)
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) % n)   # a launcher may expose one device per rank
    return _default_ctx
