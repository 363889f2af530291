"""HIP primitives vs NumPy through the C ABI (bayens_test_gemm / _potrf / _trsm)."""
import numpy as np
import pytest

from bayesian_ensembling_amd import _lib as L
from conftest import relerr

pytestmark = pytest.mark.gpu
TOL = 1e-12   # fp64 MFMA: products are exact fma chains; only the summation order differs from BLAS


@pytest.fixture(params=["tile128", "tile64"])
def gemm_kernel(request, ctx):
    """Every product case runs through BOTH fp64 product kernels: the 128 x 128-tile one (gemm_f64_mfma, what the large launches
    of a fit take) and the 64 x 64-tile one (gemm_tile64, small outputs) - by size these cases would all take the second."""
    ctx.set_gemm_tile64_max(0 if request.param == "tile128" else 1 << 30)
    yield request.param
    ctx.set_gemm_tile64_max(-1)


@pytest.mark.parametrize("nb,m,n,k", [(2, 128, 128, 64), (1, 192, 64, 128), (3, 320, 256, 192), (1, 64, 64, 64),
                                      (1, 448, 576, 64)])
@pytest.mark.parametrize("ta", [0, 1])
@pytest.mark.parametrize("tb", [0, 1])
def test_gemm_all_layouts(ctx, gemm_kernel, nb, m, n, k, ta, tb):
    rng = np.random.default_rng(nb * 1000 + m + n + k + 2 * ta + tb)
    A = rng.standard_normal((nb, k, m) if ta else (nb, m, k))
    B = rng.standard_normal((nb, n, k) if tb else (nb, k, n))
    C0 = rng.standard_normal((nb, m, n))
    flags = (L.GEMM_TA if ta else 0) | (L.GEMM_TB if tb else 0)
    out = ctx.test_gemm(A, B, C0, alpha=0.7, beta=-0.3, flags=flags, diag_add=0.25)
    ref = 0.7 * (np.swapaxes(A, 1, 2) if ta else A) @ (np.swapaxes(B, 1, 2) if tb else B) - 0.3 * C0
    idx = np.arange(min(m, n))
    ref[:, idx, idx] += 0.25
    assert relerr(out, ref) < TOL


def test_gemm_is_exact_on_integer_data_with_asymmetric_operands(ctx, gemm_kernel):
    # catches a row/col swap in the fp64 MFMA fragment maps (which differ from the f32 ones)
    i = np.arange(128.0)
    A = (i[:, None] * 3 + i[None, :64])[None]          # 128 x 64, asymmetric
    B = (i[:64, None] - 2 * i[None, :])[None]          # 64 x 128, asymmetric
    out = ctx.test_gemm(A, B, np.zeros((1, 128, 128)))
    assert np.array_equal(out, A @ B)


@pytest.mark.parametrize("nb", [2, 16, 32])
def test_gemm_triangular_output_and_ktrim(ctx, gemm_kernel, nb):
    # nb = 16 / 32: the member-interleaved tile dealing of triangular launches with nb % 8 == 0 and more than one
    # member per XCD (gemm.hip: mpc = nb >> 3 > 1) - the branch the 32-member bench and config 4's 16-member share run
    rng = np.random.default_rng(5 + nb)
    n = 320
    A = np.tril(rng.standard_normal((nb, n, n)))
    B = np.tril(rng.standard_normal((nb, n, n)))
    C0 = rng.standard_normal((nb, n, n))
    out = ctx.test_gemm(A, B, C0, flags=L.GEMM_TA | L.GEMM_TRI_OUT | L.GEMM_KTRIM_TN_LOWER)
    phi = np.swapaxes(A, 1, 2) @ B
    i = np.arange(n)
    blk = (i[:, None] // 64) >= (i[None, :] // 64)
    assert relerr(out[:, blk], phi[:, blk]) < TOL
    assert np.array_equal(out[:, ~blk], C0[:, ~blk])       # 64-blocks above the diagonal are never written
    out = ctx.test_gemm(A, B, C0, flags=L.GEMM_TA | L.GEMM_TRI_OUT | L.GEMM_KTRIM_TN_LOWER | L.GEMM_ZERO_SKIPPED)
    assert relerr(out[:, blk], phi[:, blk]) < TOL and not out[:, ~blk].any()   # ... or are zero-filled on request


@pytest.mark.parametrize("m,k,n", [(320, 320, 192), (448, 256, 128), (128, 384, 64)])
def test_gemm_lower_trapezoidal_a_stops_at_the_tiles_last_row(ctx, gemm_kernel, m, k, n):
    """GEMM_KTRIM_A_LOWER (the update product of the solve recursion with a lower-triangular right-hand side): A_ik = 0 for
    k > i, so the k loop of a 128-row tile ends at its last row - same result as the untrimmed product; and the elements
    beyond that row are really not read (NaNs planted there do not reach the result)."""
    rng = np.random.default_rng(m + k + n)
    A = np.tril(rng.standard_normal((2, m, k)))
    B = rng.standard_normal((2, k, n))
    C0 = rng.standard_normal((2, m, n))
    ref = C0 - A @ B
    out = ctx.test_gemm(A, B, C0, alpha=-1.0, beta=1.0, flags=L.GEMM_KTRIM_A_LOWER)
    assert relerr(out, ref) < TOL
    Ap = A.copy()
    i = np.arange(m)[:, None]
    Ap[:, np.broadcast_to(np.arange(k)[None, :] >= (i // 128) * 128 + 128, (m, k))] = np.nan
    out = ctx.test_gemm(Ap, B, C0, alpha=-1.0, beta=1.0, flags=L.GEMM_KTRIM_A_LOWER)
    assert relerr(out, ref) < TOL


def test_gemm_empty_k_writes_beta_c(ctx, gemm_kernel):
    # k-trimmed tile whose k range is empty cannot occur for square operands; beta path with alpha=0 instead
    rng = np.random.default_rng(6)
    A, B, C0 = rng.standard_normal((1, 64, 64)), rng.standard_normal((1, 64, 64)), rng.standard_normal((1, 64, 64))
    out = ctx.test_gemm(A, B, C0, alpha=0.0, beta=2.0)
    assert np.allclose(out, 2.0 * C0, rtol=0, atol=1e-15)


@pytest.mark.parametrize("nb,n", [(2, 64), (2, 192), (1, 512), (1, 1024)])
def test_blocked_cholesky(ctx, nb, n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((nb, n, n))
    S = M @ np.swapaxes(M, 1, 2) + n * np.eye(n)
    Lg, info = ctx.test_potrf(S)
    assert not info.any()
    assert relerr(Lg, np.linalg.cholesky(S)) < 1e-12
    assert np.array_equal(np.triu(Lg, 1), np.zeros_like(Lg))


def test_fused_256_diagonal_block_cholesky_and_its_block_inverses(ctx):
    """csrc/potrf256.hip (one launch per 256 x 256 diagonal block, the block's tiles register-resident): the factor against
    LAPACK with garbage above the diagonal on input (only the lower triangle is read), zeros above it on output, a batch larger
    than one, every pivot position reported, and - through a triangular solve that multiplies by them - the inverted 64 x 64
    diagonal blocks it leaves for the fused panel solves."""
    rng = np.random.default_rng(256)
    nb, n = 5, 512
    M = rng.standard_normal((nb, n, n))
    S = M @ np.swapaxes(M, 1, 2) + n * np.eye(n)
    Sin = np.tril(S) + np.triu(rng.standard_normal((nb, n, n)) * 1e3, 1)          # the strict upper triangle is never read
    Lg, info = ctx.test_potrf(Sin)
    ref = np.linalg.cholesky(S)
    assert not info.any() and relerr(Lg, ref) < 1e-12
    assert np.array_equal(np.triu(Lg, 1), np.zeros_like(Lg))
    for piv in (0, 15, 16, 100, 255, 256 + 77, 511):
        bad = np.eye(n)[None].repeat(2, axis=0)
        bad[1, piv, piv] = -1.0
        _, info = ctx.test_potrf(bad)
        assert info[0] == 0 and info[1] == piv + 1, piv
    Bm = rng.standard_normal((nb, 64, n))
    want = np.stack([np.linalg.solve(ref[b], Bm[b].T).T for b in range(nb)])      # B L^-T
    assert relerr(ctx.test_trsm(ref, Bm, 0), want) < 1e-11


def test_cholesky_reports_the_failing_pivot(ctx):
    S = np.eye(256)[None].repeat(2, axis=0)
    S[1, 130, 130] = -3.0
    _, info = ctx.test_potrf(S)
    assert info[0] == 0 and info[1] == 131


@pytest.mark.parametrize("nb,m,n", [(2, 128, 192), (1, 256, 256), (1, 64, 448)])
def test_triangular_solves(ctx, nb, m, n):
    rng = np.random.default_rng(m + n)
    M = rng.standard_normal((nb, n, n))
    Lt = np.linalg.cholesky(M @ np.swapaxes(M, 1, 2) + n * np.eye(n))
    Bm = rng.standard_normal((nb, m, n))
    r0 = np.stack([np.linalg.solve(Lt[b], Bm[b].T).T for b in range(nb)])
    r1 = np.stack([np.linalg.solve(Lt[b].T, Bm[b].T).T for b in range(nb)])
    assert relerr(ctx.test_trsm(Lt, Bm, 0), r0) < 1e-11
    assert relerr(ctx.test_trsm(Lt, Bm, 1), r1) < 1e-11


def test_triangular_solve_with_lower_rhs(ctx):
    rng = np.random.default_rng(9)
    n = 320
    M = rng.standard_normal((1, n, n))
    Lt = np.linalg.cholesky(M @ np.swapaxes(M, 1, 2) + n * np.eye(n))
    Bm = np.tril(rng.standard_normal((1, n, n)))
    want = np.linalg.solve(Lt[0].T, Bm[0].T).T
    got = ctx.test_trsm(Lt, Bm, 2)[0]
    assert relerr(np.tril(got), np.tril(want)) < 1e-11
