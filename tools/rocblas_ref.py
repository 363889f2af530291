"""Reference point for the GEMM kernel: the vendor library's batched fp64 GEMM (rocblas_dgemm_strided_batched through
ctypes) on the largest product shapes of the fit, on the same device buffers libbayens allocates.  Measurement only -
nothing in the package links rocBLAS."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bayesian_ensembling_amd import _lib as L

os.environ["BAYENS_BENCH_BETA"] = "1"       # bayens_bench_gemm accumulates into C, like the rocBLAS call below
ctx = L.Context(0)
rb = C.CDLL("/opt/rocm/lib/librocblas.so")
handle = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(handle)) == 0
rb.rocblas_dgemm_strided_batched.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(C.c_double), C.c_void_p, C.c_int, C.c_longlong,
                                             C.c_void_p, C.c_int, C.c_longlong, C.POINTER(C.c_double),
                                             C.c_void_p, C.c_int, C.c_longlong, C.c_int]
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
NONE, TRANS = 111, 112
nb = 32
for (m, n, k, ta, tb, label) in [(2048, 1024, 1024, NONE, TRANS, "NT 2048x1024x1024"), (2048, 1024, 1024, NONE, NONE, "NN 2048x1024x1024"),
                                 (2048, 2048, 2048, NONE, TRANS, "NT 2048^3"), (2048, 256, 256, NONE, TRANS, "NT 2048x256x256"),
                                 (1536, 1536, 256, NONE, TRANS, "NT 1536x1536x256")]:
    # row-major C = A op(B) is column-major C^T = op(B)^T A^T: swap the operands and the m / n roles
    A = ctx.empty((nb, m, k)); B = ctx.empty((nb, n, k) if tb == TRANS else (nb, k, n)); Cm = ctx.empty((nb, m, n))
    alpha, beta = C.c_double(1.0), C.c_double(1.0)
    ldb = k if tb == TRANS else n

    def run():
        rc = rb.rocblas_dgemm_strided_batched(handle, tb, NONE, n, m, k, C.byref(alpha), B.ptr, ldb, n * k, A.ptr, k, m * k,
                                              C.byref(beta), Cm.ptr, n, m * n, nb)
        assert rc == 0, rc
    run(); hip.hipDeviceSynchronize()
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    hip.hipDeviceSynchronize()
    dt = (time.perf_counter() - t0) / reps
    ms_b = ctx.bench_gemm(nb, m, n, k, (L.GEMM_TB if tb == TRANS else 0), reps=10)
    print(f"{label:22s} rocBLAS {dt * 1e3:8.3f} ms  {2.0 * m * n * k * nb / dt / 1e12:6.2f} TFLOP/s   libbayens {ms_b:8.3f} ms  {2.0 * m * n * k * nb / (ms_b * 1e-3) / 1e12:6.2f} TFLOP/s", flush=True)
    for d in (A, B, Cm):
        d.free()
