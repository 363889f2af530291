// fit_f32.hip - fp32 instance of the fit (BASELINE.json config 3).  This translation unit, gemm.hip, kernels.hip and
// driver.cpp are compiled a second time with -DBAYENS_F32: `real` = float, namespace bayens_f32, the 16x16x4 MFMA
// in its fp32 form (same operand layout, twice the rate, half the bytes in every O(N^2) pass).  What stays fp64 in
// this mode: the hyper-parameters and Adam state, the squared distances of the kernel matrix (X~X~^T on the fp64
// MFMA - r^2 = |x|^2 + |y|^2 - 2 x.y cancels), the 64x64 diagonal-block Cholesky + inverse (LDS images and MFMA in
// fp64, loaded from / stored to fp32), and every reduction (log-determinants, traces, gradient sums, the loss).
#ifndef BAYENS_F32
#error "compile with -DBAYENS_F32"
#endif
#include "fit_f32.h"

#include <cstdlib>
#include <vector>

#include "fit_ws.h"
#include "small.h"

namespace bayens_f32 {
struct State {
  hipStream_t stream;
  HipPrims prims;
  DeviceWs dws;
  int* small_info = nullptr;       // device info[] of the fused small-member path
  int small_info_cap = 0, n_cus = 256;
  double* kpark = nullptr;         // kernel-value park of the fused small-member kernel
  size_t kpark_cap = 0;
  State(hipStream_t s, hipStream_t side) : stream(s), prims(s) {
    prims.borrow_side_stream(side);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      n_cus = prop.multiProcessorCount;
  }
  ~State() { if (small_info) (void)hipFree(small_info); if (kpark) (void)hipFree(kpark); }
};
}  // namespace bayens_f32

using namespace bayens_f32;

void* bayens_f32_create(hipStream_t stream, hipStream_t side) { return new State(stream, side); }
void bayens_f32_destroy(void* state) { delete static_cast<State*>(state); }

void bayens_f32_profile_enable(void* state, int on) { static_cast<State*>(state)->prims.profile_reset(on != 0); }
void bayens_f32_profile_add(void* state, int family, double* ms, double* flop, long* launches, double* bytes) {
  HipPrims& p = static_cast<State*>(state)->prims;
  if (ms) *ms += p.profile_ms(family);
  if (flop) *flop += p.prof.flop[family];
  if (launches) *launches += p.prof.launches[family];
  if (bytes) *bytes += p.prof.bytes[family];
}

int bayens_f32_fit(void* state, const BayensF32Args& a, std::string* err) {
  State* st = static_cast<State*>(state);
  try {
    static const bool no_small = getenv("BAYENS_NO_SMALL") != nullptr;
    if (!no_small && smalln::small_fit_supported(a.N, a.R)) {      // small members: the fused one-kernel-per-fit path (small.hip)
      if (st->small_info_cap < a.B) {
        if (st->small_info) (void)hipFree(st->small_info);
        st->small_info = nullptr;
        BAYENS_HIP(hipMalloc((void**)&st->small_info, (size_t)a.B * sizeof(int)));
        st->small_info_cap = a.B;
      }
      smalln::SmallArgs g{a.B, a.N, a.R, a.kernel, a.n_iters, a.gamma, a.lr, a.jitter, a.X, a.y_mean, a.y_var, a.mu, a.cov,
                          a.var, a.chol, a.logdet, a.hyp, a.basis, st->small_info};
      g.loss_stride = (a.n_iters + LOSS_EVERY - 1) / LOSS_EVERY;
      g.loss = g.loss_stride > 0 ? a.loss_trace : nullptr;
      const size_t need = smalln::small_park_doubles(a.B, a.N, st->n_cus);
      if (need > st->kpark_cap) {
        if (st->kpark) (void)hipFree(st->kpark);
        st->kpark = nullptr;
        BAYENS_HIP(hipMalloc((void**)&st->kpark, need * sizeof(double)));
        st->kpark_cap = need;
      }
      g.kpark = st->kpark;
      g.Rv = a.Rv;
      g.xoff = a.xoff;
      smalln::small_fit(st->stream, g, st->n_cus);
      BAYENS_HIP(hipMemcpyAsync(a.info_host, st->small_info, (size_t)a.B * sizeof(int), hipMemcpyDeviceToHost, st->stream));
      BAYENS_HIP(hipStreamSynchronize(st->stream));
      return 0;
    }
    const int Np = pad_to_leaf(a.N);
    st->dws.ensure(a.B, Np);
    FitWorkspace w = st->dws.view(0, a.B);
    FitParams fp{a.N, a.R, a.kernel, a.n_iters, a.gamma, a.lr, a.jitter};
    fp.loss_stride = (a.n_iters + LOSS_EVERY - 1) / LOSS_EVERY;
    fp.loss_trace = fp.loss_stride > 0 ? a.loss_trace : nullptr;
    fp.Rv = a.Rv;
    fp.xoff = a.xoff;
    gp_fit(st->prims, w, fp, a.X, a.y_mean, a.y_var, a.mu, a.cov, a.chol, a.logdet, a.hyp, a.basis, a.var);
    BAYENS_HIP(hipMemcpyAsync(a.info_host, w.info, (size_t)a.B * sizeof(int), hipMemcpyDeviceToHost, st->stream));
    BAYENS_HIP(hipStreamSynchronize(st->stream));
    return 0;
  } catch (const HipError& e) {
    if (err) *err = e.what;
    return e.code > 0 ? e.code : -1;
  }
}
