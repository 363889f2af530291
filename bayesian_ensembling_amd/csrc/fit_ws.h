// fit_ws.h - device workspace of the batched fit (allocation, member-range views).  Written in `real`, so it
// serves the fp64 context (capi.hip) and the fp32 one (fit_f32.hip, compiled with -DBAYENS_F32).
#pragma once
#include <cstdlib>
#include <vector>

#include "driver.h"
#include "hip_prims.h"

namespace BAYENS_NS {

struct DeviceWs {
  FitWorkspace ws;
  std::vector<void*> allocs;

  ~DeviceWs() { release(); }
  void release() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    ws = FitWorkspace();
  }
  // six [nb][Np][Np] matrices (T1 with 64 extra rows; Lb and Cb interleaved), two sets of inverted diagonal blocks, the vectors
  void ensure(int nb, int Np) {
    if (ws.nb >= nb && ws.Np == Np) return;
    release();
    auto dalloc = [&](size_t bytes) {
      void* p = nullptr;
      BAYENS_HIP(hipMalloc(&p, bytes));
      allocs.push_back(p);
      return p;
    };
    FitWorkspace& w = ws;
    w.nb = nb; w.Np = Np;
    const size_t msz = (size_t)nb * Np * Np * sizeof(real);
    auto mat = [&]() { return Mat{(real*)dalloc(msz), Np, (long)Np * Np}; };
    w.W0 = mat(); w.Lam = mat(); w.T2 = mat();
    // Lb and Cb interleaved member by member (driver.h FitWorkspace::paired)
    static const bool flat = std::getenv("BAYENS_NO_INTERLEAVE") != nullptr;   // (measurement switch)
    if (flat) {
      w.Lb = mat(); w.Cb = mat();
    } else {
      real* lc = (real*)dalloc(2 * msz);
      w.Lb = Mat{lc, Np, 2L * Np * Np};
      w.Cb = Mat{lc + (long)Np * Np, Np, 2L * Np * Np};
    }
    w.T1 = Mat{(real*)dalloc((size_t)nb * (Np + LEAF) * Np * sizeof(real)), Np, (long)(Np + LEAF) * Np};
    const size_t isz = (size_t)nb * Np * LEAF * sizeof(real);
    if (flat) { w.invL = (real*)dalloc(isz); w.invC = (real*)dalloc(isz); w.inv_step = 0; }
    else { w.invL = (real*)dalloc(2 * isz); w.invC = w.invL + (long)Np * LEAF; w.inv_step = 2L * Np * LEAF; }
    const size_t vsz = (size_t)nb * Np * sizeof(real);
    w.theta1 = (real*)dalloc(vsz); w.mvec = (real*)dalloc(vsz); w.avec = (real*)dalloc(vsz);
    w.y = (real*)dalloc(vsz); w.nu = (real*)dalloc(vsz); w.muvec = (real*)dalloc(vsz);
    w.uvec = (real*)dalloc(vsz);
    w.ustate = (double*)dalloc((size_t)nb * 8 * sizeof(double));
    w.dscratch = (double*)dalloc((size_t)nb * 4 * sizeof(double));
    w.info = (int*)dalloc((size_t)nb * sizeof(int));
    w.info_pair = (int*)dalloc((size_t)2 * nb * sizeof(int));
    BAYENS_HIP(hipMemset(w.info_pair, 0, (size_t)2 * nb * sizeof(int)));   // (info_fold leaves it cleared)
  }
  // a workspace view for members [b0, b0 + nb) of the allocated batch
  FitWorkspace view(int b0, int nb) const {
    FitWorkspace v = ws;
    v.nb = nb;
    for (Mat* m : {&v.W0, &v.Lb, &v.Lam, &v.Cb, &v.T1, &v.T2}) m->p += (long)b0 * m->stride;
    const long is = v.inv_stride();
    v.invL += b0 * is; v.invC += b0 * is;
    for (real** p : {&v.theta1, &v.mvec, &v.avec, &v.y, &v.nu, &v.muvec, &v.uvec}) *p += (long)b0 * v.Np;
    v.ustate += (long)b0 * 8;
    v.dscratch += (long)b0 * 4;
    v.info += b0;
    v.info_pair += 2 * b0;
    return v;
  }
};

}  // namespace BAYENS_NS
