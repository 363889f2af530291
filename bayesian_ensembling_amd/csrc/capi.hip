// capi.hip - the extern "C" boundary of libbayens (include/bayens.h): context, HBM plumbing, the hot-path
// entry points and the RCCL all-gather.  No torch types, no exceptions across the boundary.
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/bayens.h"
#include "../../include/bayens_diag.h"
#include "driver.h"
#include "fit_f32.h"
#include "fit_ws.h"
#include "hip_prims.h"
#include "medium.h"
#include "potrf256.h"
#include "small.h"

using namespace bayens;

// ---- minimal RCCL surface, resolved with dlopen so that the library loads on hosts without RCCL ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
typedef ncclResult_t (*pfn_ncclGetUniqueId)(ncclUniqueId*);
typedef ncclResult_t (*pfn_ncclCommInitRank)(ncclComm_t*, int, ncclUniqueId, int);
typedef ncclResult_t (*pfn_ncclCommDestroy)(ncclComm_t);
typedef ncclResult_t (*pfn_ncclAllGather)(const void*, void*, size_t, int /*ncclDataType_t*/, ncclComm_t, hipStream_t);
typedef ncclResult_t (*pfn_ncclAllReduce)(const void*, void*, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, ncclComm_t,
                                          hipStream_t);
typedef const char* (*pfn_ncclGetErrorString)(ncclResult_t);
constexpr int NCCL_FLOAT64 = 8;
constexpr int NCCL_SUM = 0, NCCL_MAX = 2;

struct RcclApi {
  void* handle = nullptr;
  std::string path;
  pfn_ncclGetUniqueId GetUniqueId = nullptr;
  pfn_ncclCommInitRank CommInitRank = nullptr;
  pfn_ncclCommDestroy CommDestroy = nullptr;
  pfn_ncclAllGather AllGather = nullptr;
  pfn_ncclAllReduce AllReduce = nullptr;
  pfn_ncclGetErrorString GetErrorString = nullptr;
  bool load(std::string& err) {
    if (handle) return true;
    // ROCm's own RCCL first, by absolute path (ROCM_PATH, then /opt/rocm): it is built against the libamdhip64 this
    // library links.  The bare sonames come last - under them the loader would hand back whatever copy a host
    // program (e.g. PyTorch's bundled one) happens to have mapped already.
    std::vector<std::string> names;
    if (const char* rp = getenv("ROCM_PATH")) names.push_back(std::string(rp) + "/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so");
    names.push_back("librccl.so.1");
    names.push_back("librccl.so");
    for (const auto& n : names) {
      handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (handle) { path = n; break; }
    }
    if (!handle) { err = std::string("dlopen(librccl): ") + dlerror(); return false; }
    GetUniqueId = (pfn_ncclGetUniqueId)dlsym(handle, "ncclGetUniqueId");
    CommInitRank = (pfn_ncclCommInitRank)dlsym(handle, "ncclCommInitRank");
    CommDestroy = (pfn_ncclCommDestroy)dlsym(handle, "ncclCommDestroy");
    AllGather = (pfn_ncclAllGather)dlsym(handle, "ncclAllGather");
    AllReduce = (pfn_ncclAllReduce)dlsym(handle, "ncclAllReduce");
    GetErrorString = (pfn_ncclGetErrorString)dlsym(handle, "ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !AllGather) { err = "librccl: missing symbols"; return false; }
    return true;
  }
};

struct bayens_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  HipPrims* prims = nullptr;
  std::string err;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  // fit workspace (grown on demand)
  DeviceWs dws;
  void* f32 = nullptr;             // fp32 instance of the fit (fit_f32.hip), created on first use
  // member groups: the batch of a fit is split over `gstreams` HIP streams (one host thread each) so that the
  // low-parallelism steps of one group (64x64 diagonal blocks, narrow panels) overlap the big GEMMs of the others
  std::vector<hipStream_t> gstreams;
  std::vector<HipPrims*> gprims;
  int ngroups = 0;                 // member groups of a fit batch; 0 = by batch size (fit_in_groups)
  double* sscratch = nullptr;      // small persistent device scratch (per-member scalars)
  size_t sscratch_cap = 0;
  double* small_scratch(size_t count) {
    if (count > sscratch_cap) {
      if (sscratch) (void)hipFree(sscratch);
      sscratch = nullptr;
      if (hipMalloc((void**)&sscratch, count * sizeof(double)) != hipSuccess) throw HipError{BAYENS_E_ALLOC, "hipMalloc"};
      sscratch_cap = count;
    }
    return sscratch;
  }
  int n_cus = 256;                 // compute units of the device (grid sizing of the fused small-member kernel)
  int medium_min_batch = 0;        // > 0: batches of at least this many mid-size members take the fused kernel (medium.hip); 0: by size
  int fit_route = 0;               // 0: fused kernels by size (default); 1: general path whatever the size (bayens_set_fit_route, fp64)
  long chain_stalls = 0;           // bayens_loglik_rows calls whose chained substitution gave up and took the blocked solve
  double* mscratch = nullptr;      // tile scratch of the fused mid-size kernel (medium.hip), grown on demand
  size_t mscratch_cap = 0;
  double* medium_scratch(size_t count) {
    if (count > mscratch_cap) {
      if (mscratch) (void)hipFree(mscratch);
      mscratch = nullptr;
      if (hipMalloc((void**)&mscratch, count * sizeof(double)) != hipSuccess) throw HipError{BAYENS_E_ALLOC, "hipMalloc"};
      mscratch_cap = count;
    }
    return mscratch;
  }
  unsigned long long* small_stamps = nullptr;   // diagnostic phase stamps of the fused kernel (bayens_small_stamps)
  double* kpark = nullptr;         // kernel-value park of the fused small-member kernel (small.hip), grown on demand
  size_t kpark_cap = 0;
  double* small_park(size_t count) {
    if (count > kpark_cap) {
      if (kpark) (void)hipFree(kpark);
      kpark = nullptr;
      if (hipMalloc((void**)&kpark, count * sizeof(double)) != hipSuccess) throw HipError{BAYENS_E_ALLOC, "hipMalloc"};
      kpark_cap = count;
    }
    return kpark;
  }
  RcclApi rccl;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  double* xbuf = nullptr;          // packed rows of the member exchange: [rows][3][P] sent | [nranks][rows][3][P] received
  size_t xbuf_cap = 0;
  double* exchange_scratch(size_t count) {
    if (count > xbuf_cap) {
      if (xbuf) (void)hipFree(xbuf);
      xbuf = nullptr;
      if (hipMalloc((void**)&xbuf, count * sizeof(double)) != hipSuccess) throw HipError{BAYENS_E_ALLOC, "hipMalloc"};
      xbuf_cap = count;
    }
    return xbuf;
  }
};

namespace {

int fail(bayens_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

template <typename F>
int guarded(bayens_ctx* c, F&& f) {
  if (!c) return BAYENS_E_ARG;
  try {
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, (int)e, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return f();
  } catch (const HipError& e) {
    return fail(c, e.code != 0 ? e.code : BAYENS_E_ARG, e.what);   // hipError_t (> 0) or a BAYENS_E_* code (< 0) as thrown
  } catch (const std::exception& e) {
    return fail(c, BAYENS_E_ALLOC, e.what());
  }
}

void free_ws(bayens_ctx* c) { c->dws.release(); }
void ensure_ws(bayens_ctx* c, int nb, int Np) { c->dws.ensure(nb, Np); }
FitWorkspace ws_view(bayens_ctx* c, int b0, int nb) { return c->dws.view(b0, nb); }

// Runs the fit of B resident members split into groups, one HIP stream + host thread per group.
void fit_in_groups(bayens_ctx* c, int B, const FitParams& fp, const double* dX, const double* dy, const double* dnu,
                   double* dmu, double* dcov, double* dchol, double* dld, double* dhyp, double* dbasis, double* dvar = nullptr) {
  const int N = fp.N, R = fp.R;
  // One group by default: with a group's stream and its look-ahead stream on different hardware queues (bayens_ctx_create) the
  // trailing updates really run beside the next diagonal block and panel solve, and one group of 32 members is the fastest
  // arrangement - 57.7 ms per 32 x 2000 step against 58.6 with two groups on four queues (one half's latency-bound steps beside
  // the other half's products) and 69 with four (profiles/r05_hw_queues.txt).  bayens_set_groups(ctx, n > 0) overrides.
  const int want = c->ngroups > 0 ? c->ngroups : 1;
  const int G = std::max(1, std::min(std::min(want, (int)c->gstreams.size()), B));
  BAYENS_HIP(hipStreamSynchronize(c->stream));       // staged inputs are in HBM before any group starts
  if (G == 1) {                                      // one group: no host thread, run on the first group stream
    FitWorkspace w = ws_view(c, 0, B);
    gp_fit(*c->gprims[0], w, fp, dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis, dvar);
    BAYENS_HIP(hipStreamSynchronize(c->gstreams[0]));
    return;
  }
  std::vector<std::thread> th;
  std::vector<HipError> errs(G, HipError{0, ""});
  for (int g = 0; g < G; ++g) {
    const int b0 = (int)((long)B * g / G), b1 = (int)((long)B * (g + 1) / G);
    if (b1 <= b0) continue;
    th.emplace_back([=, &errs]() {
      try {
        BAYENS_HIP(hipSetDevice(c->device));
        FitWorkspace w = ws_view(c, b0, b1 - b0);
        FitParams fg = fp;
        if (fg.loss_trace) fg.loss_trace += (size_t)b0 * fg.loss_stride;
        if (fg.Rv) { fg.Rv += b0; fg.xoff += b0; }          // (a ragged batch: the offsets are absolute, X's base stays)
        gp_fit(*c->gprims[g], w, fg, fg.Rv ? dX : dX + (size_t)b0 * N * R, dy + (size_t)b0 * N, dnu + (size_t)b0 * N,
               dmu + (size_t)b0 * N, dcov ? dcov + (size_t)b0 * N * N : nullptr, dchol ? dchol + (size_t)b0 * N * N : nullptr,
               dld ? dld + b0 : nullptr, dhyp ? dhyp + (size_t)b0 * 2 : nullptr,
               dbasis ? dbasis + (size_t)b0 * 2 * N : nullptr, dvar ? dvar + (size_t)b0 * N : nullptr);
        BAYENS_HIP(hipStreamSynchronize(c->gstreams[g]));
      } catch (const HipError& e) {
        errs[g] = e;
      }
    });
  }
  for (auto& t : th) t.join();
  for (auto& e : errs)
    if (e.code != 0) throw e;
}

// host staging helpers ------------------------------------------------------------------------
struct Stage {
  bayens_ctx* c;
  int loc;
  std::vector<void*> owned;
  struct Out { void* host; void* dev; size_t bytes; };
  std::vector<Out> outs;
  Stage(bayens_ctx* c_, int loc_) : c(c_), loc(loc_) {}
  ~Stage() { for (void* p : owned) (void)hipFree(p); }
  template <typename T>
  const T* in(const T* p, size_t count) {
    if (!p || loc == BAYENS_DEVICE) return p;
    void* d = nullptr;
    BAYENS_HIP(hipMalloc(&d, count * sizeof(T)));
    owned.push_back(d);
    BAYENS_HIP(hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return (const T*)d;
  }
  template <typename T>
  T* out(T* p, size_t count) {
    if (!p || loc == BAYENS_DEVICE) return p;
    void* d = nullptr;
    BAYENS_HIP(hipMalloc(&d, count * sizeof(T)));
    owned.push_back(d);
    outs.push_back(Out{p, d, count * sizeof(T)});
    return (T*)d;
  }
  template <typename T>
  T* scratch(size_t count) {
    void* d = nullptr;
    BAYENS_HIP(hipMalloc(&d, count * sizeof(T)));
    owned.push_back(d);
    return (T*)d;
  }
  void finish() {
    for (auto& o : outs) BAYENS_HIP(hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
  }
};

}  // namespace

extern "C" {

const char* bayens_version(void) { return "bayens 0.1 (gfx950, fp64 + fp32 MFMA)"; }

int bayens_device_count(int* count) {
  if (!count) return BAYENS_E_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = e == hipSuccess ? n : 0;
  return e == hipSuccess ? 0 : BAYENS_E_NOGPU;
}

int bayens_ctx_create(int device, bayens_ctx** out) {
  if (!out) return BAYENS_E_ARG;
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return BAYENS_E_NOGPU;
  if (device < 0 || device >= ndev) return BAYENS_E_ARG;
  e = hipSetDevice(device);
  if (e != hipSuccess) return (int)e;
  bayens_ctx* c = new bayens_ctx();
  c->device = device;
  e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return (int)e; }
  (void)hipEventCreate(&c->t0);
  (void)hipEventCreate(&c->t1);
  c->prims = new HipPrims(c->stream);
  try {
    c->prims->ensure_side_stream();          // (its look-ahead stream right behind it: see the group streams below)
  } catch (const HipError& err) {
    bayens_ctx_destroy(c);
    return err.code ? err.code : BAYENS_E_INTERNAL;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) c->n_cus = prop.multiProcessorCount;
  }
  if (const char* env = getenv("BAYENS_STREAMS")) c->ngroups = std::max(1, std::min(16, atoi(env)));
  // The HIP runtime deals its streams over a handful of hardware queues (four by default) in the order they are created, and
  // kernels of two streams only run beside each other when the streams sit on DIFFERENT queues.  Round 5 measured what that does
  // to a 32 x 2000 step when the look-ahead streams were created on first use, by whichever host thread got there first: 58.7 ms
  // in most processes, 60.4 in some, and the one-group batch 60.1 or 57.7 - the same kernels (profiles/r05_hw_queues.txt).  So
  // the streams a fit uses are created HERE, in a fixed order: group g's stream, then its look-ahead stream, group by group - a
  // group's two streams never share a queue, and neither do those of groups 0 and 1.
  if (const char* sk = getenv("BAYENS_STREAM_SKEW")) {      // (measurement switch: shifts which hardware queues the fit's streams get)
    for (int k = 0; k < atoi(sk); ++k) { hipStream_t d = nullptr; (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking); }
  }
  for (int g = 0; g < 16; ++g) {
    hipStream_t s = nullptr;
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { bayens_ctx_destroy(c); return (int)e; }
    c->gstreams.push_back(s);
    c->gprims.push_back(new HipPrims(s));
    try {
      c->gprims.back()->ensure_side_stream();
    } catch (const HipError& err) {
      bayens_ctx_destroy(c);
      return err.code ? err.code : BAYENS_E_INTERNAL;
    }
  }
  *out = c;
  return 0;
}

int bayens_ctx_destroy(bayens_ctx* c) {
  if (!c) return BAYENS_E_ARG;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm && c->rccl.CommDestroy) c->rccl.CommDestroy(c->comm);
  free_ws(c);
  if (c->f32) bayens_f32_destroy(c->f32);
  if (c->sscratch) (void)hipFree(c->sscratch);
  if (c->mscratch) (void)hipFree(c->mscratch);
  if (c->xbuf) (void)hipFree(c->xbuf);
  if (c->small_stamps) (void)hipFree(c->small_stamps);
  if (c->kpark) (void)hipFree(c->kpark);
  for (auto* p : c->gprims) delete p;
  for (auto s : c->gstreams) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
  delete c->prims;
  if (c->t0) (void)hipEventDestroy(c->t0);
  if (c->t1) (void)hipEventDestroy(c->t1);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return 0;
}

const char* bayens_last_error(bayens_ctx* c) { return c ? c->err.c_str() : "null context"; }

int bayens_device_info(bayens_ctx* c, char* name, int name_len, int* cus, size_t* hbm) {
  return guarded(c, [&]() {
    hipDeviceProp_t p;
    BAYENS_HIP(hipGetDeviceProperties(&p, c->device));
    if (name && name_len > 0) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
    if (cus) *cus = p.multiProcessorCount;
    if (hbm) *hbm = p.totalGlobalMem;
    return 0;
  });
}

int bayens_malloc(bayens_ctx* c, size_t bytes, void** dptr) {
  return guarded(c, [&]() { BAYENS_HIP(hipMalloc(dptr, bytes)); return 0; });
}
int bayens_free(bayens_ctx* c, void* dptr) {
  return guarded(c, [&]() { BAYENS_HIP(hipFree(dptr)); return 0; });
}
int bayens_memcpy_h2d(bayens_ctx* c, void* dst, const void* src, size_t bytes) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    return 0;
  });
}
int bayens_memcpy_d2h(bayens_ctx* c, void* dst, const void* src, size_t bytes) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    return 0;
  });
}
int bayens_memcpy_d2d(bayens_ctx* c, void* dst, const void* src, size_t bytes) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    return 0;
  });
}
int bayens_sync(bayens_ctx* c) {
  return guarded(c, [&]() { BAYENS_HIP(hipStreamSynchronize(c->stream)); return 0; });
}
int bayens_timer_start(bayens_ctx* c) {
  return guarded(c, [&]() { BAYENS_HIP(hipEventRecord(c->t0, c->stream)); return 0; });
}
int bayens_timer_stop(bayens_ctx* c, double* ms) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipEventRecord(c->t1, c->stream));
    BAYENS_HIP(hipEventSynchronize(c->t1));
    float f = 0;
    BAYENS_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    if (ms) *ms = f;
    return 0;
  });
}
int bayens_profile_enable(bayens_ctx* c, int on) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipDeviceSynchronize());
    std::vector<HipPrims*> all = c->gprims;
    all.push_back(c->prims);
    for (HipPrims* hp : all) hp->profile_reset(on != 0);
    if (on && !c->f32) c->f32 = bayens_f32_create(c->stream, c->prims->side_stream);
    if (c->f32) bayens_f32_profile_enable(c->f32, on);
    return 0;
  });
}
int bayens_profile_bytes(bayens_ctx* c, int family, double* bytes) {
  return guarded(c, [&]() {
    if (family < 0 || family >= FAM_COUNT || !bytes) return fail(c, BAYENS_E_ARG, "unknown kernel family");
    std::vector<HipPrims*> all = c->gprims;
    all.push_back(c->prims);
    double b = 0;
    for (HipPrims* hp : all) b += hp->prof.bytes[family];
    if (c->f32) bayens_f32_profile_add(c->f32, family, nullptr, nullptr, nullptr, &b);
    *bytes = b;
    return 0;
  });
}
int bayens_profile_get(bayens_ctx* c, int family, double* ms, double* flop, long* launches) {
  return guarded(c, [&]() {
    if (family < 0 || family >= FAM_COUNT) return fail(c, BAYENS_E_ARG, "unknown kernel family");
    BAYENS_HIP(hipDeviceSynchronize());
    std::vector<HipPrims*> all = c->gprims;
    all.push_back(c->prims);
    double tot = 0, fl = 0;
    long n = 0;
    for (HipPrims* hp : all) {
      tot += hp->profile_ms(family);
      fl += hp->prof.flop[family];
      n += hp->prof.launches[family];
    }
    if (c->f32) bayens_f32_profile_add(c->f32, family, &tot, &fl, &n, nullptr);
    if (ms) *ms = tot;
    if (flop) *flop = fl;
    if (launches) *launches = n;
    return 0;
  });
}

int bayens_profile_stages(bayens_ctx* c, char* buf, int buf_len) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipDeviceSynchronize());
    auto& marks = c->gprims[0]->prof.marks;
    std::string out;
    std::vector<std::pair<std::string, double>> agg;
    for (size_t i = 0; i + 1 < marks.size(); ++i) {
      float f = 0;
      BAYENS_HIP(hipEventElapsedTime(&f, marks[i].second, marks[i + 1].second));
      bool found = false;
      for (auto& a : agg) if (a.first == marks[i].first) { a.second += f; found = true; }
      if (!found) agg.emplace_back(marks[i].first, f);
    }
    char line[160];
    for (auto& a : agg) { snprintf(line, sizeof line, "%-24s %9.3f ms\n", a.first.c_str(), a.second); out += line; }
    for (auto& m : marks) (void)hipEventDestroy(m.second);
    marks.clear();
    if (buf && buf_len > 0) snprintf(buf, buf_len, "%s", out.c_str());
    return 0;
  });
}

int bayens_profile_shapes(bayens_ctx* c, char* buf, int buf_len) {
  return guarded(c, [&]() {
    BAYENS_HIP(hipDeviceSynchronize());
    struct Agg { int fam, m, n, k, flags; long calls; double ms, flop; };
    std::vector<Agg> agg;
    std::vector<HipPrims*> all = c->gprims;
    all.push_back(c->prims);
    for (HipPrims* hp : all)
      for (auto& e : hp->prof.events) {
        float f = 0;
        BAYENS_HIP(hipEventElapsedTime(&f, e.a, e.b));
        bool found = false;
        for (auto& a : agg)
          if (a.fam == e.fam && a.m == e.m && a.n == e.n && a.k == e.k && a.flags == e.flags) {
            a.calls++; a.ms += f; a.flop += e.flop; found = true; break;
          }
        if (!found) agg.push_back(Agg{e.fam, e.m, e.n, e.k, e.flags, 1, f, e.flop});
      }
    std::string out = "fam     m     n     k flags calls        ms   TFLOP/s\n";
    char line[160];
    for (auto& a : agg) {
      snprintf(line, sizeof line, "%3d %5d %5d %5d %5d %5ld %9.3f %9.2f\n", a.fam, a.m, a.n, a.k, a.flags, a.calls, a.ms,
               a.flop / (a.ms * 1e-3) / 1e12);
      out += line;
    }
    if (buf && buf_len > 0) snprintf(buf, buf_len, "%s", out.c_str());
    return 0;
  });
}

int bayens_set_groups(bayens_ctx* c, int n_groups) {
  if (!c || n_groups < 0 || n_groups > 16) return BAYENS_E_ARG;
  c->ngroups = n_groups;
  return 0;
}

int bayens_set_medium_min_batch(bayens_ctx* c, int min_batch) {
  if (!c || min_batch < 0) return BAYENS_E_ARG;
  c->medium_min_batch = min_batch;
  return 0;
}

int bayens_chain_stalls(bayens_ctx* c, long* count) {
  if (!c || !count) return BAYENS_E_ARG;
  *count = c->chain_stalls;
  return 0;
}

int bayens_set_gemm_tile64_max(bayens_ctx* c, int max_workgroups) {
  if (!c || max_workgroups < -1) return BAYENS_E_ARG;
  c->prims->tile64_max = max_workgroups;
  for (HipPrims* hp : c->gprims) hp->tile64_max = max_workgroups;
  return 0;
}

int bayens_set_fit_route(bayens_ctx* c, int route) {
  if (!c || route < 0 || route > 1) return BAYENS_E_ARG;
  c->fit_route = route;
  return 0;
}

}  // extern "C"

namespace {
// bayens_gp_fit_batch / bayens_gp_fit_batch_ragged.  Rv / x_off (HOST arrays, null: every member has R realisations): member b
// has Rv[b] <= R realisations, its [N][Rv[b]] block starts x_off[b] doubles into X, which holds x_total doubles in all.
int fit_batch_impl(bayens_ctx* c, const char* who, int loc, int B, int N, int R, const int* Rv, const size_t* x_off, size_t x_total,
                   const double* X, const double* y_mean, const double* y_var, int kernel, int n_iters, double gamma, double lr,
                   double jitter, double* mu, double* cov, double* cov_chol, double* logdet, double* hyp, double* loss_trace,
                   int* info, double* ll_basis) {
  const std::string tag = std::string(who) + ": ";
  // small members (N <= 94): the whole fit is ONE kernel, one workgroup per member, matrices resident in LDS (small.hip)
  static const bool no_small = getenv("BAYENS_NO_SMALL") != nullptr;
  const bool small = !no_small && c->fit_route == 0 && smalln::small_fit_supported(N, R);
  // mid-size members (94 < N <= 254, the reference's N = 165 series): ONE kernel too, matrices in an L2-resident scratch
  // (medium.hip; BAYENS_NO_MEDIUM=1 forces the general path)
  // Measured (profiles/r03_medium_kernel.txt, us per optimiser iteration): the fused kernel is flat in the batch - one CU per
  // member: 124 at N = 100, 322 at N = 165, 858 at N = 250 up to 256 members - where the general path, which shares every
  // launch between the members, goes 228 / 357 / 488 at 20 members and 717 at (N = 165, 256 members).  So: up to 12 tiles
  // per side (N <= 190) the fused kernel always; above that from 128 members on (bayens_set_medium_min_batch overrides).
  static const bool no_medium = getenv("BAYENS_NO_MEDIUM") != nullptr;
  const int medium_from = c->medium_min_batch > 0 ? c->medium_min_batch : ((N + 2 + 15) / 16 <= 12 ? 1 : 128);
  const bool medium = !small && !no_medium && c->fit_route == 0 && B >= medium_from && mediumn::medium_fit_supported(N, R);
  const int Np = pad_to_leaf(N);
  if (!small && !medium) ensure_ws(c, B, Np);
  Stage st(c, loc);
  const double* dX = st.in(X, x_total);
  const double* dy = st.in(y_mean, (size_t)B * N);
  const double* dnu = st.in(y_var, (size_t)B * N);
  double* dmu = st.out(mu, (size_t)B * N);
  double* dcov = st.out(cov, (size_t)B * N * N);
  double* dchol = st.out(cov_chol, (size_t)B * N * N);
  double* dld = st.out(logdet, (size_t)B);
  double* dhyp = st.out(hyp, (size_t)B * 2);
  double* dbasis = st.out(ll_basis, (size_t)B * 2 * N);
  // the ragged layout goes to the device as two small arrays (they are host arrays whatever `loc` says)
  const int* dRv = nullptr;
  const long* dxoff = nullptr;
  std::vector<long> hoff;
  if (Rv) {
    int* r = st.scratch<int>((size_t)B);
    long* o = st.scratch<long>((size_t)B);
    hoff.resize(B);
    for (int b = 0; b < B; ++b) hoff[b] = (long)x_off[b];
    BAYENS_HIP(hipMemcpyAsync(r, Rv, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipMemcpyAsync(o, hoff.data(), (size_t)B * sizeof(long), hipMemcpyHostToDevice, c->stream));
    dRv = r;
    dxoff = o;
  }
  auto report = [&](const int* dinfo) {
    std::vector<int> hinfo(B, 0);
    BAYENS_HIP(hipMemcpyAsync(hinfo.data(), dinfo, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    int bad = 0;
    for (int b = 0; b < B; ++b) { if (info) info[b] = hinfo[b]; if (hinfo[b]) bad = hinfo[b]; }
    if (bad) return fail(c, BAYENS_E_NOTPD, tag + "matrix not positive definite (see info[])");
    return 0;
  };
  const int loss_stride = (n_iters + LOSS_EVERY - 1) / LOSS_EVERY;
  if (small) {
    int* dinfo = st.scratch<int>((size_t)B);
    smalln::SmallArgs a{B, N, R, kernel, n_iters, gamma, lr, jitter, dX, dy, dnu, dmu, dcov, nullptr, dchol, dld, dhyp, dbasis, dinfo};
    a.stamps = c->small_stamps;
    a.kpark = c->small_park(smalln::small_park_doubles(B, N, c->n_cus));
    a.loss_stride = loss_stride;
    a.loss = (loss_trace && a.loss_stride > 0) ? st.out(loss_trace, (size_t)B * a.loss_stride) : nullptr;
    a.Rv = dRv;
    a.xoff = dxoff;
    smalln::small_fit(c->stream, a, c->n_cus);
    return report(dinfo);
  }
  if (medium) {
    int* dinfo = st.scratch<int>((size_t)B);
    mediumn::MediumArgs a{B, N, R, kernel, n_iters, gamma, lr, jitter, dX, dy, dnu, dmu, dcov, nullptr, dchol, dld, dhyp, dbasis, dinfo,
                          c->medium_scratch((size_t)mediumn::medium_grid(B, N, c->n_cus) * mediumn::medium_scratch_doubles(N))};
    a.stamps = c->small_stamps;
    a.loss_stride = loss_stride;
    a.loss = (loss_trace && a.loss_stride > 0) ? st.out(loss_trace, (size_t)B * a.loss_stride) : nullptr;
    a.Rv = dRv;
    a.xoff = dxoff;
    a.x_elems = (long)x_total;
    mediumn::medium_fit(c->stream, a, c->n_cus);
    return report(dinfo);
  }
  FitParams fp{N, R, kernel, n_iters, gamma, lr, jitter};
  fp.loss_stride = loss_stride;
  fp.loss_trace = fp.loss_stride > 0 ? st.out(loss_trace, (size_t)B * fp.loss_stride) : nullptr;
  fp.Rv = dRv;
  fp.xoff = dxoff;
  fit_in_groups(c, B, fp, dX, dy, dnu, dmu, dcov, dchol, dld, dhyp, dbasis);
  return report(c->dws.ws.info);
}

// bayens_gp_fit_batch_f32 / bayens_gp_fit_batch_ragged_f32 (Rv / x_off as in fit_batch_impl)
int fit_batch_f32_impl(bayens_ctx* c, const char* who, int loc, int B, int N, int R, const int* Rv, const size_t* x_off, size_t x_total,
                       const float* X, const float* y_mean, const float* y_var, int kernel, int n_iters, double gamma, double lr,
                       double jitter, float* mu, float* cov, float* cov_chol, double* logdet, double* hyp, double* loss_trace,
                       int* info, float* ll_basis) {
  const std::string tag = std::string(who) + ": ";
  if (!c->f32) c->f32 = bayens_f32_create(c->stream, c->prims->side_stream);
  Stage st(c, loc);
  BayensF32Args a;
  a.B = B; a.N = N; a.R = R; a.kernel = kernel; a.n_iters = n_iters;
  a.gamma = gamma; a.lr = lr; a.jitter = jitter;
  a.X = st.in(X, x_total);
  a.y_mean = st.in(y_mean, (size_t)B * N);
  a.y_var = st.in(y_var, (size_t)B * N);
  a.mu = st.out(mu, (size_t)B * N);
  a.cov = st.out(cov, (size_t)B * N * N);
  a.chol = st.out(cov_chol, (size_t)B * N * N);
  a.logdet = st.out(logdet, (size_t)B);
  a.hyp = st.out(hyp, (size_t)B * 2);
  const int ntrace = (n_iters + LOSS_EVERY - 1) / LOSS_EVERY;
  a.loss_trace = ntrace > 0 ? st.out(loss_trace, (size_t)B * ntrace) : nullptr;
  a.basis = st.out(ll_basis, (size_t)B * 2 * N);
  std::vector<long> hoff;
  if (Rv) {
    int* r = st.scratch<int>((size_t)B);
    long* o = st.scratch<long>((size_t)B);
    hoff.resize(B);
    for (int b = 0; b < B; ++b) hoff[b] = (long)x_off[b];
    BAYENS_HIP(hipMemcpyAsync(r, Rv, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipMemcpyAsync(o, hoff.data(), (size_t)B * sizeof(long), hipMemcpyHostToDevice, c->stream));
    a.Rv = r;
    a.xoff = o;
  }
  std::vector<int> hinfo(B, 0);
  a.info_host = hinfo.data();
  std::string err;
  const int rc = bayens_f32_fit(c->f32, a, &err);
  if (rc != 0) return fail(c, rc > 0 ? rc : BAYENS_E_ARG, tag + err);
  st.finish();
  int bad = 0;
  for (int b = 0; b < B; ++b) { if (info) info[b] = hinfo[b]; if (hinfo[b]) bad = hinfo[b]; }
  if (bad) return fail(c, BAYENS_E_NOTPD, tag + "matrix not positive definite (see info[])");
  return 0;
}
}  // namespace

extern "C" {

int bayens_gp_fit_batch(bayens_ctx* c, int loc, int B, int N, int R, const double* X, const double* y_mean,
                        const double* y_var, int kernel, int n_iters, double gamma, double lr, double jitter,
                        double* mu, double* cov, double* cov_chol, double* logdet, double* hyp,
                        double* loss_trace, int* info, double* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R <= 0 || !X || !y_mean || !y_var || !mu || !cov || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch: unknown kernel id");
    return fit_batch_impl(c, "gp_fit_batch", loc, B, N, R, nullptr, nullptr, (size_t)B * N * R, X, y_mean, y_var, kernel, n_iters,
                          gamma, lr, jitter, mu, cov, cov_chol, logdet, hyp, loss_trace, info, ll_basis);
  });
}

int bayens_gp_fit_batch_ragged(bayens_ctx* c, int loc, int B, int N, const int* R, const size_t* x_off, const double* X,
                               const double* y_mean, const double* y_var, int kernel, int n_iters, double gamma, double lr,
                               double jitter, double* mu, double* cov, double* cov_chol, double* logdet, double* hyp,
                               double* loss_trace, int* info, double* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || !R || !X || !y_mean || !y_var || !mu || !cov || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged: unknown kernel id");
    int Rmax = 0;
    std::vector<size_t> off(B);
    size_t total = 0;
    for (int b = 0; b < B; ++b) {
      if (R[b] <= 0) return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged: every member needs at least one realisation");
      Rmax = std::max(Rmax, R[b]);
      off[b] = x_off ? x_off[b] : total;                 // null: the members' blocks lie one behind the other
      total = std::max(total, off[b] + (size_t)N * R[b]);
    }
    return fit_batch_impl(c, "gp_fit_batch_ragged", loc, B, N, Rmax, R, off.data(), total, X, y_mean, y_var, kernel, n_iters, gamma,
                          lr, jitter, mu, cov, cov_chol, logdet, hyp, loss_trace, info, ll_basis);
  });
}

int bayens_transpose_rows(bayens_ctx* c, int loc, int nb, int A, int B, const double* src, double* dst) {
  return guarded(c, [&]() {
    if (nb <= 0 || A <= 0 || B <= 0 || nb > 65535 || !src || !dst || src == dst)
      return fail(c, BAYENS_E_ARG, "transpose_rows: bad argument");
    Stage st(c, loc);
    const double* ds = st.in(src, (size_t)nb * A * B);
    double* dd = st.out(dst, (size_t)nb * A * B);
    c->prims->transpose_rows(nb, A, B, ds, dd);
    st.finish();
    return 0;
  });
}

int bayens_field_prologue(bayens_ctx* c, int loc, int M, int R, int N, int G, const double* raw, double* X, double* y_mean,
                          double* y_var) {
  return guarded(c, [&]() {
    if (M <= 0 || R <= 0 || N <= 0 || G <= 0 || M > 65535 || N > 65535 || !raw || !X || !y_mean || !y_var)
      return fail(c, BAYENS_E_ARG, "field_prologue: bad argument");
    Stage st(c, loc);
    const double* draw = st.in(raw, (size_t)M * R * N * G);
    double* dX = st.out(X, (size_t)M * G * N * R);
    double* dy = st.out(y_mean, (size_t)M * G * N);
    double* dnu = st.out(y_var, (size_t)M * G * N);
    c->prims->field_prologue(M, R, N, G, draw, dX, dy, dnu);
    st.finish();
    return 0;
  });
}

int bayens_gp_fit_batch_reduced(bayens_ctx* c, int loc, int B, int N, int R, const double* X, const double* y_mean,
                                const double* y_var, int kernel, int n_iters, double gamma, double lr, double jitter,
                                double* mu, double* var, double* logdet, double* hyp, int* info, double* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R <= 0 || !X || !y_mean || !y_var || !mu || !var || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch_reduced: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch_reduced: unknown kernel id");
    const bool medium = !smalln::small_fit_supported(N, R) && mediumn::medium_fit_supported(N, R);
    const bool general = !smalln::small_fit_supported(N, R) && !medium;
    if (general) ensure_ws(c, B, pad_to_leaf(N));
    Stage st(c, loc);
    const double* dX = st.in(X, (size_t)B * N * R);
    const double* dy = st.in(y_mean, (size_t)B * N);
    const double* dnu = st.in(y_var, (size_t)B * N);
    double* dmu = st.out(mu, (size_t)B * N);
    double* dvar = st.out(var, (size_t)B * N);
    double* dld = st.out(logdet, (size_t)B);
    double* dhyp = st.out(hyp, (size_t)B * 2);
    double* dbasis = st.out(ll_basis, (size_t)B * 2 * N);
    int* dinfo = general ? c->dws.ws.info : st.scratch<int>((size_t)B);
    if (general) {
      // N > 254 (the reference's monthly series, N = 1032 / 1980: experiments/data): the general path, whose covariance stays in
      // the workspace - its diagonal, log-determinant and log-likelihood basis are what leaves
      FitParams fp{N, R, kernel, n_iters, gamma, lr, jitter};
      fit_in_groups(c, B, fp, dX, dy, dnu, dmu, nullptr, nullptr, dld, dhyp, dbasis, dvar);
    } else if (medium) {
      // 94 < N <= 254 (the reference's historical annual horizon, N = 165: experiments/full_experiment_script.py:57): the fused
      // mid-size kernel with the same reduced export - no covariance, no factor leaves the kernel
      mediumn::MediumArgs a{B, N, R, kernel, n_iters, gamma, lr, jitter, dX, dy, dnu, dmu, nullptr, dvar, nullptr, dld, dhyp, dbasis, dinfo,
                            c->medium_scratch((size_t)mediumn::medium_grid(B, N, c->n_cus) * mediumn::medium_scratch_doubles(N))};
      a.stamps = c->small_stamps;
      mediumn::medium_fit(c->stream, a, c->n_cus);
    } else {
      smalln::SmallArgs a{B, N, R, kernel, n_iters, gamma, lr, jitter, dX, dy, dnu, dmu, nullptr, dvar, nullptr, dld, dhyp, dbasis, dinfo};
      a.stamps = c->small_stamps;
      a.kpark = c->small_park(smalln::small_park_doubles(B, N, c->n_cus));
      smalln::small_fit(c->stream, a, c->n_cus);
    }
    std::vector<int> hinfo(B, 0);
    BAYENS_HIP(hipMemcpyAsync(hinfo.data(), dinfo, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    int bad = 0;
    for (int b = 0; b < B; ++b) { if (info) info[b] = hinfo[b]; if (hinfo[b]) bad = hinfo[b]; }
    if (bad) return fail(c, BAYENS_E_NOTPD, "gp_fit_batch_reduced: matrix not positive definite (see info[])");
    return 0;
  });
}

int bayens_gp_fit_batch_f32(bayens_ctx* c, int loc, int B, int N, int R, const float* X, const float* y_mean,
                            const float* y_var, int kernel, int n_iters, double gamma, double lr, double jitter,
                            float* mu, float* cov, float* cov_chol, double* logdet, double* hyp, double* loss_trace,
                            int* info, float* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R <= 0 || !X || !y_mean || !y_var || !mu || !cov || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch_f32: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch_f32: unknown kernel id");
    return fit_batch_f32_impl(c, "gp_fit_batch_f32", loc, B, N, R, nullptr, nullptr, (size_t)B * N * R, X, y_mean, y_var, kernel, n_iters,
                              gamma, lr, jitter, mu, cov, cov_chol, logdet, hyp, loss_trace, info, ll_basis);
  });
}

int bayens_gp_fit_batch_ragged_f32(bayens_ctx* c, int loc, int B, int N, const int* R, const size_t* x_off, const float* X,
                                   const float* y_mean, const float* y_var, int kernel, int n_iters, double gamma, double lr,
                                   double jitter, float* mu, float* cov, float* cov_chol, double* logdet, double* hyp,
                                   double* loss_trace, int* info, float* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || !R || !X || !y_mean || !y_var || !mu || !cov || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged_f32: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged_f32: unknown kernel id");
    int Rmax = 0;
    std::vector<size_t> off(B);
    size_t total = 0;
    for (int b = 0; b < B; ++b) {
      if (R[b] <= 0) return fail(c, BAYENS_E_ARG, "gp_fit_batch_ragged_f32: every member needs at least one realisation");
      Rmax = std::max(Rmax, R[b]);
      off[b] = x_off ? x_off[b] : total;
      total = std::max(total, off[b] + (size_t)N * R[b]);
    }
    return fit_batch_f32_impl(c, "gp_fit_batch_ragged_f32", loc, B, N, Rmax, R, off.data(), total, X, y_mean, y_var, kernel, n_iters,
                              gamma, lr, jitter, mu, cov, cov_chol, logdet, hyp, loss_trace, info, ll_basis);
  });
}

int bayens_gp_fit_batch_reduced_f32(bayens_ctx* c, int loc, int B, int N, int R, const float* X, const float* y_mean,
                                    const float* y_var, int kernel, int n_iters, double gamma, double lr, double jitter,
                                    float* mu, float* var, double* logdet, double* hyp, int* info, float* ll_basis) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R <= 0 || !X || !y_mean || !y_var || !mu || !var || n_iters < 0)
      return fail(c, BAYENS_E_ARG, "gp_fit_batch_reduced_f32: bad argument");
    if (kernel < 0 || kernel > 2) return fail(c, BAYENS_E_ARG, "gp_fit_batch_reduced_f32: unknown kernel id");
    if (!c->f32) c->f32 = bayens_f32_create(c->stream, c->prims->side_stream);
    Stage st(c, loc);
    BayensF32Args a;
    a.B = B; a.N = N; a.R = R; a.kernel = kernel; a.n_iters = n_iters;
    a.gamma = gamma; a.lr = lr; a.jitter = jitter;
    a.X = st.in(X, (size_t)B * N * R);
    a.y_mean = st.in(y_mean, (size_t)B * N);
    a.y_var = st.in(y_var, (size_t)B * N);
    a.mu = st.out(mu, (size_t)B * N);
    a.cov = nullptr;
    a.var = st.out(var, (size_t)B * N);
    a.chol = nullptr;
    a.logdet = st.out(logdet, (size_t)B);
    a.hyp = st.out(hyp, (size_t)B * 2);
    a.loss_trace = nullptr;
    a.basis = st.out(ll_basis, (size_t)B * 2 * N);
    std::vector<int> hinfo(B, 0);
    a.info_host = hinfo.data();
    std::string err;
    const int rc = bayens_f32_fit(c->f32, a, &err);
    if (rc != 0) return fail(c, rc > 0 ? rc : BAYENS_E_ARG, "gp_fit_batch_reduced_f32: " + err);
    st.finish();
    int bad = 0;
    for (int b = 0; b < B; ++b) { if (info) info[b] = hinfo[b]; if (hinfo[b]) bad = hinfo[b]; }
    if (bad) return fail(c, BAYENS_E_NOTPD, "gp_fit_batch_reduced_f32: matrix not positive definite (see info[])");
    return 0;
  });
}

int bayens_mvn_factor_batch(bayens_ctx* c, int loc, int B, int N, const double* cov, const double* mu, double* chol,
                            double* logdet, double* ll_basis, int* info) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || !cov) return fail(c, BAYENS_E_ARG, "mvn_factor_batch: bad argument");
    const int Np = pad_to_leaf(N);
    ensure_ws(c, B, Np);
    Stage st(c, loc);
    const double* dcov = st.in(cov, (size_t)B * N * N);
    double* dchol = st.out(chol, (size_t)B * N * N);
    double* dld = st.out(logdet, (size_t)B);
    if (ll_basis && !mu) return fail(c, BAYENS_E_ARG, "mvn_factor_batch: ll_basis needs mu");
    const double* dmu = st.in(mu, (size_t)B * N);
    double* dbasis = st.out(ll_basis, (size_t)B * 2 * N);
    FitWorkspace w = ws_view(c, 0, B);
    BAYENS_HIP(hipMemsetAsync(w.info, 0, (size_t)B * sizeof(int), c->stream));
    c->prims->pad_in(B, N, Np, dcov, w.T2);
    if (dbasis) c->prims->pad_vec(B, N, Np, dmu, w.muvec);
    mvn_factor(*c->prims, w, N, w.muvec, dchol, dld, dbasis);
    std::vector<int> hinfo(B, 0);
    BAYENS_HIP(hipMemcpyAsync(hinfo.data(), w.info, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    int bad = 0;
    for (int b = 0; b < B; ++b) { if (info) info[b] = hinfo[b]; if (hinfo[b]) bad = hinfo[b]; }
    if (bad) return fail(c, BAYENS_E_NOTPD, "mvn_factor_batch: covariance not positive definite (see info[])");
    return 0;
  });
}

int bayens_extract_diag(bayens_ctx* c, int loc, int B, int N, const double* cov, double* var) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || !cov || !var) return fail(c, BAYENS_E_ARG, "extract_diag: bad argument");
    Stage st(c, loc);
    const double* dcov = st.in(cov, (size_t)B * N * N);
    double* dvar = st.out(var, (size_t)B * N);
    c->prims->extract_diag(B, N, dcov, dvar);
    st.finish();
    return 0;
  });
}

// fp32 posteriors feed the (fp64) weight and scheme kernels through these two widening copies: the [M][N] tables
// are a few MB, the covariance itself is never widened
namespace {
__global__ void widen_f32_kernel(size_t n, const float* src, double* dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}
__global__ void extract_diag_f32_kernel(int N, const float* cov, double* var) {
  const int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) var[(size_t)b * N + i] = (double)cov[((size_t)b * N + i) * N + i];
}
}  // namespace

__global__ void narrow_f64_kernel(size_t n, const double* src, float* dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

int bayens_narrow_f64(bayens_ctx* c, int loc, size_t n, const double* src, float* dst) {
  return guarded(c, [&]() {
    if (n == 0 || !src || !dst) return fail(c, BAYENS_E_ARG, "narrow_f64: bad argument");
    Stage st(c, loc);
    const double* ds = st.in(src, n);
    float* dd = st.out(dst, n);
    hipLaunchKernelGGL(narrow_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, ds, dd);
    BAYENS_HIP(hipGetLastError());
    st.finish();
    return 0;
  });
}

int bayens_widen_f32(bayens_ctx* c, int loc, size_t n, const float* src, double* dst) {
  return guarded(c, [&]() {
    if (n == 0 || !src || !dst) return fail(c, BAYENS_E_ARG, "widen_f32: bad argument");
    Stage st(c, loc);
    const float* ds = st.in(src, n);
    double* dd = st.out(dst, n);
    hipLaunchKernelGGL(widen_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, n, ds, dd);
    BAYENS_HIP(hipGetLastError());
    st.finish();
    return 0;
  });
}

int bayens_extract_diag_f32(bayens_ctx* c, int loc, int B, int N, const float* cov, double* var) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || !cov || !var) return fail(c, BAYENS_E_ARG, "extract_diag_f32: bad argument");
    Stage st(c, loc);
    const float* dcov = st.in(cov, (size_t)B * N * N);
    double* dvar = st.out(var, (size_t)B * N);
    hipLaunchKernelGGL(extract_diag_f32_kernel, dim3((N + 255) / 256, B), dim3(256), 0, c->stream, N, dcov, dvar);
    BAYENS_HIP(hipGetLastError());
    st.finish();
    return 0;
  });
}

namespace {
// rows <- rows L^-T for a caller-supplied compact lower factor (device): the factor is padded into the fit workspace
// (lower 64-blocks only), its 64x64 diagonal blocks are inverted in one launch, and the rows ride through the blocked
// right-side solve of the fit (trsm_rlt) - one pass over the factor at the GEMM's pace instead of a serial substitution.
// Returns the 64-row right-hand-side block (ws.T1's first rows) for the caller to fill BEFORE calling `solve`.
struct FactorSolve {
  bayens_ctx* c;
  FitWorkspace w;
  int B, N, Np;
  FactorSolve(bayens_ctx* c_, int B_, int N_, const double* dchol) : c(c_), B(B_), N(N_), Np(pad_to_leaf(N_)) {
    ensure_ws(c, B, Np);
    w = ws_view(c, 0, B);
    BAYENS_HIP(hipMemsetAsync(w.info, 0, (size_t)B * sizeof(int), c->stream));
    c->prims->pad_lower(B, N, Np, dchol, w.T2);
    c->prims->tri_inv64(B, Np, w.T2, w.invL, w.inv_stride(), w.info);
  }
  // first member whose factor has a diagonal entry that is not a positive finite number (0 = none); one host wait
  int bad_member(int* pivot) {
    std::vector<int> h(B);
    BAYENS_HIP(hipMemcpyAsync(h.data(), w.info, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    for (int b = 0; b < B; ++b)
      if (h[b]) { *pivot = h[b]; return b + 1; }
    return 0;
  }
  Mat rows() const { return w.T1; }
  void solve() { trsm_rlt(*c->prims, B, LEAF, Np, w.T2, w.invL, w.inv_stride(), w.T1); }
};

// The two-right-hand-side case (the log-likelihood basis L^-1 1, L^-1 mu) straight from the COMPACT factor, in one
// launch and one pass over its lower triangle: workgroup (J, m) owns the 64 rows J of member m.  It inverts its own
// diagonal block in LDS and forms M = L_JJ^-1 L_J,J-1, then walks the blocks k < J - 1 of its block row (fetched a
// block ahead; the factor does not depend on the chain), multiplying each by z_k as soon as workgroup (k, m) has
// published it, prepares v = L_JJ^-1 (b_J - sum_{k<J-1} L_Jk z_k), and publishes
//   z_J = v - M z_J-1
// - the only work between seeing z_J-1 and publishing z_J is one 64 x 64 product done by ONE wave per right-hand side
// from registers (lane = row, z_J-1 broadcast lane by lane with v_readlane): no LDS, no barrier on the chain.
// Publication is the data itself: the exchange buffer starts as all-ones words (a NaN no arithmetic produces) and a
// consumer polls the word it needs with device-scope atomic loads, so there is no separate flag round trip.  Block
// ids are J-major: a workgroup only ever waits for workgroups with smaller ids, which the dispatcher has started
// before it, so the chain cannot deadlock however many workgroups are resident; a poll that sees nothing for
// CHAIN_POLL_LIMIT tries gives up (err = 1) instead of spinning for ever.
constexpr int CHAIN_B = 64, CHAIN_LS = 65, CHAIN_POLL_LIMIT = 1 << 21;
constexpr unsigned long long CHAIN_EMPTY = ~0ull;

__device__ __forceinline__ double chain_readlane(double v, int src_lane) {
  const unsigned lo = __builtin_amdgcn_readlane((int)__double2loint(v), src_lane);
  const unsigned hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double((int)hi, (int)lo);
}

// polls one published word; returns false when the producer never showed up
__device__ __forceinline__ bool chain_poll(const double* src, double& v) {
  const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long bits = CHAIN_EMPTY;
  for (int tries = 0; tries < CHAIN_POLL_LIMIT; ++tries) {
    bits = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bits != CHAIN_EMPTY) break;
    __builtin_amdgcn_s_sleep(1);
  }
  v = __longlong_as_double((long long)bits);
  return bits != CHAIN_EMPTY;
}

__global__ __launch_bounds__(256) void fwd2_chain_kernel(int B, int N, int nblk, const double* __restrict__ chol,
                                                         const double* __restrict__ mu, double* zbuf,
                                                         double* __restrict__ basis, int* err) {
  __shared__ double Ds[CHAIN_B * CHAIN_LS];      // the diagonal block, its inverse, then M
  __shared__ double Ls[CHAIN_B * CHAIN_LS];      // the off-diagonal block being multiplied
  __shared__ double zs[2 * CHAIN_B];
  __shared__ double red[4 * 2 * CHAIN_B];
  __shared__ int dead;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int J = blockIdx.x / B, m = blockIdx.x - J * B;
  const int r0 = J * CHAIN_B;
  const double* Lm = chol + (size_t)m * N * N;
  double* zm = zbuf + (size_t)m * nblk * 2 * CHAIN_B;
  double regs[16];
  // rows r0 + 16 wave + i, column 64 k + lane of the factor: zero above the diagonal, identity in the padding
  auto load_block = [&](int k) {
    const int col = k * CHAIN_B + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = r0 + 16 * wave + i;
      double v = 0.0;
      if (row < N && col <= row) v = Lm[(size_t)row * N + col];
      else if (row >= N && col == row) v = 1.0;
      regs[i] = v;
    }
  };
  auto store_block = [&](double* dst) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dst[(16 * wave + i) * CHAIN_LS + lane] = regs[i];
  };
  // red <- partial sums over this wave's 16 columns of (coefficients a[i]) x (the two vectors in zs); lane = row
  auto partial = [&](const double* a) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      s0 = fma(a[i], zs[16 * wave + i], s0);
      s1 = fma(a[i], zs[CHAIN_B + 16 * wave + i], s1);
    }
    red[(wave * 2 + 0) * CHAIN_B + lane] = s0;
    red[(wave * 2 + 1) * CHAIN_B + lane] = s1;
  };
  auto reduced = [&]() {   // threads < 128: vector `wave`, row `lane`
    return (red[(0 * 2 + wave) * CHAIN_B + lane] + red[(1 * 2 + wave) * CHAIN_B + lane]) +
           (red[(2 * 2 + wave) * CHAIN_B + lane] + red[(3 * 2 + wave) * CHAIN_B + lane]);
  };

  if (t == 0) dead = 0;
  load_block(J);
  store_block(Ds);
  if (J > 0) load_block(J - 1);                  // in flight while the diagonal block is inverted
  __syncthreads();
  if (wave == 0) {
    // X = L_JJ^-1 in place, column `lane`: X_i = (e_i - sum_{k<i} L_ik X_k) / L_ii.  Row i of L is read before it is
    // overwritten (one wave, program order); X_k = 0 above the diagonal, so every lane runs the same loop.
    for (int i = 0; i < CHAIN_B; ++i) {
      double s0 = 0.0, s1 = 0.0;
      int k = 0;
      for (; k + 1 < i; k += 2) {
        s0 = fma(Ds[i * CHAIN_LS + k], Ds[k * CHAIN_LS + lane], s0);
        s1 = fma(Ds[i * CHAIN_LS + k + 1], Ds[(k + 1) * CHAIN_LS + lane], s1);
      }
      if (k < i) s0 = fma(Ds[i * CHAIN_LS + k], Ds[k * CHAIN_LS + lane], s0);
      const double d = Ds[i * CHAIN_LS + i];
      Ds[i * CHAIN_LS + lane] = ((i == lane ? 1.0 : 0.0) - (s0 + s1)) / d;
    }
  }
  if (J > 0) store_block(Ls);                    // L_J,J-1
  __syncthreads();
  double dinv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dinv[i] = Ds[lane * CHAIN_LS + 16 * wave + i];
  if (J > 0) {
    if (J > 1) load_block(0);
    // M = L_JJ^-1 L_J,J-1: row `lane`, this wave's 16 columns (L_JJ^-1 is lower triangular: k <= lane)
    double mrow[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mrow[i] = 0.0;
    for (int k = 0; k < CHAIN_B; ++k) {
      const double d = Ds[lane * CHAIN_LS + k];
#pragma unroll
      for (int i = 0; i < 16; ++i) mrow[i] = fma(d, Ls[k * CHAIN_LS + 16 * wave + i], mrow[i]);
    }
    __syncthreads();                             // every wave has read the inverse: M replaces it
#pragma unroll
    for (int i = 0; i < 16; ++i) Ds[lane * CHAIN_LS + 16 * wave + i] = mrow[i];
  }

  double acc = 0.0;
  for (int k = 0; k + 1 < J; ++k) {
    __syncthreads();                             // Ls, zs and red of the previous block are done with
    store_block(Ls);
    if (k + 2 < J) load_block(k + 1);
    if (t < 128) {
      double v;
      if (!chain_poll(zm + (size_t)(k * 2 + wave) * CHAIN_B + lane, v)) dead = 1;
      zs[wave * CHAIN_B + lane] = v;
    }
    __syncthreads();
    if (dead) {
      if (t == 0) *err = 1;
      return;
    }
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = Ls[lane * CHAIN_LS + 16 * wave + i];
    partial(a);
    __syncthreads();
    if (t < 128) acc += reduced();
  }
  __syncthreads();
  if (t < 128) {
    const int row = r0 + lane;
    const double b = row < N ? (wave == 0 ? 1.0 : mu[(size_t)m * N + row]) : 0.0;
    zs[wave * CHAIN_B + lane] = b - acc;
  }
  __syncthreads();
  partial(dinv);
  __syncthreads();
  if (t >= 128) return;
  // the chain step: one wave per right-hand side, registers only
  double z = reduced();
  if (J > 0) {
    double mfull[CHAIN_B];
#pragma unroll
    for (int c = 0; c < CHAIN_B; ++c) mfull[c] = Ds[lane * CHAIN_LS + c];
    double zp;
    const bool ok = chain_poll(zm + (size_t)((J - 1) * 2 + wave) * CHAIN_B + lane, zp);
    if (!__all(ok)) {
      if (lane == 0) *err = 1;
      return;
    }
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int c = 0; c < CHAIN_B; c += 4) {
      s0 = fma(mfull[c], chain_readlane(zp, c), s0);
      s1 = fma(mfull[c + 1], chain_readlane(zp, c + 1), s1);
      s2 = fma(mfull[c + 2], chain_readlane(zp, c + 2), s2);
      s3 = fma(mfull[c + 3], chain_readlane(zp, c + 3), s3);
    }
    z -= (s0 + s1) + (s2 + s3);
  }
  const int row = r0 + lane;
  unsigned long long* p = reinterpret_cast<unsigned long long*>(zm + (size_t)(J * 2 + wave) * CHAIN_B + lane);
  unsigned long long bits = (unsigned long long)__double_as_longlong(z);
  if (bits == CHAIN_EMPTY) bits = 0x7ff8000000000000ull;     // (an input NaN with that payload: publish a plain NaN)
  __hip_atomic_store(p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (row < N) basis[((size_t)m * 2 + wave) * N + row] = z;
}
}  // namespace

int bayens_loglik_rows(bayens_ctx* c, int loc, int B, int N, int R_obs, const double* mu, const double* cov_chol,
                       const double* logdet, const double* obs, double* ll_mean) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R_obs <= 0 || !mu || !cov_chol || !logdet || !obs || !ll_mean)
      return fail(c, BAYENS_E_ARG, "loglik_rows: bad argument");
    Stage st(c, loc);
    const double* dmu = st.in(mu, (size_t)B * N);
    const double* dch = st.in(cov_chol, (size_t)B * N * N);
    const double* dld = st.in(logdet, (size_t)B);
    const double* dobs = st.in(obs, (size_t)R_obs * N);
    double* dll = st.out(ll_mean, (size_t)B * N);
    // the basis (L^-1 1, L^-1 mu) from the compact factor in one chained launch, then the basis path
    const int nblk = (N + CHAIN_B - 1) / CHAIN_B;
    const size_t n_z = (size_t)B * nblk * 2 * CHAIN_B;
    double* dots = c->small_scratch((size_t)B * 3 + 4 + (size_t)B * 2 * N + n_z);
    double* basis = dots + (size_t)B * 3 + 4;
    double* zbuf = basis + (size_t)B * 2 * N;
    int* derr = reinterpret_cast<int*>(dots + (size_t)B * 3 + 2);
    BAYENS_HIP(hipMemsetAsync(derr, 0, sizeof(int), c->stream));
    BAYENS_HIP(hipMemsetAsync(zbuf, 0xFF, n_z * sizeof(double), c->stream));
    hipLaunchKernelGGL(fwd2_chain_kernel, dim3((unsigned)(nblk * B)), dim3(256), 0, c->stream, B, N, nblk, dch, dmu, zbuf, basis,
                       derr);
    BAYENS_HIP(hipGetLastError());
    c->prims->loglik_rows_basis(B, N, R_obs, basis, dld, dobs, dll, dots);
    int herr = 0;
    BAYENS_HIP(hipMemcpyAsync(&herr, derr, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    static const bool force_stall = getenv("BAYENS_FORCE_CHAIN_STALL") != nullptr;     // (test hook for the fallback)
    if (herr || force_stall) {
      // (the event leaves a trace: a per-context counter - bayens_chain_stalls - and, once per process, a line on stderr; a
      //  timing of this entry that took the fallback is a timing of the blocked solve, and bench.py says so)
      c->chain_stalls += 1;
      static bool told = false;
      if (!told && !force_stall) {
        told = true;
        fprintf(stderr, "libbayens: bayens_loglik_rows: the chained substitution ran out of polls (a consumer workgroup waited "
                        "for a block that was not published within 2^21 tries) - recomputed through the blocked solve; "
                        "bayens_chain_stalls() counts these events\n");
      }
      // The chain's liveness rests on producer workgroups being dispatched before their consumers and is bounded by a
      // poll count, not by progress: on a GPU shared with other streams / processes a consumer can run out of polls
      // for perfectly valid input.  That is a scheduling accident, not an error of the caller: recompute the basis
      // through the blocked solve, which has no cross-workgroup waits (pad the factor, invert its diagonal blocks,
      // carry the rows (1, mu) through trsm_rlt) - a few launches and a second pass over the factor instead of one.
      FactorSolve fs(c, B, N, dch);
      int pivot = 0;
      if (const int bad = fs.bad_member(&pivot))
        return fail(c, BAYENS_E_NOTPD, "loglik_rows: the factor of member " + std::to_string(bad - 1) +
                                           " has a diagonal entry that is not positive and finite (row " +
                                           std::to_string(pivot) + ")");
      c->prims->solve_rows_set(B, N, fs.Np, 2, 1, nullptr, 0, dmu, fs.rows());
      fs.solve();
      c->prims->basis_rows_get(B, N, fs.Np, fs.rows(), basis);
      c->prims->loglik_rows_basis(B, N, R_obs, basis, dld, dobs, dll, dots);
    }
    st.finish();
    return 0;
  });
}

int bayens_mvn_log_prob(bayens_ctx* c, int loc, int B, int N, int n_x, int shared_x, const double* mu, const double* cov_chol,
                        const double* logdet, const double* x, double* log_prob) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || n_x <= 0 || !mu || !cov_chol || !logdet || !x || !log_prob)
      return fail(c, BAYENS_E_ARG, "mvn_log_prob: bad argument");
    Stage st(c, loc);
    const double* dmu = st.in(mu, (size_t)B * N);
    const double* dch = st.in(cov_chol, (size_t)B * N * N);
    const double* dld = st.in(logdet, (size_t)B);
    const double* dx = st.in(x, (size_t)(shared_x ? 1 : B) * n_x * N);
    double* dout = st.out(log_prob, (size_t)B * n_x);
    FactorSolve fs(c, B, N, dch);
    {
      int pivot = 0;
      if (const int bad = fs.bad_member(&pivot))
        return fail(c, BAYENS_E_NOTPD, "mvn_log_prob: the factor of member " + std::to_string(bad - 1) +
                                           " has a diagonal entry that is not positive and finite (row " +
                                           std::to_string(pivot) + ")");
    }
    double* tmp = st.scratch<double>((size_t)B * LEAF);
    for (int r0 = 0; r0 < n_x; r0 += LEAF) {           // 64 arguments per pass of the factor
      const int nr = n_x - r0 < LEAF ? n_x - r0 : LEAF;
      c->prims->solve_rows_set(B, N, fs.Np, nr, 0, dx + (size_t)r0 * N, shared_x ? 0 : (long)n_x * N, dmu, fs.rows());
      fs.solve();
      c->prims->rows_logprob(B, N, nr, fs.rows(), dld, tmp);
      for (int b = 0; b < B; ++b)
        BAYENS_HIP(hipMemcpyAsync(dout + (size_t)b * n_x + r0, tmp + (size_t)b * nr, (size_t)nr * sizeof(double),
                                  hipMemcpyDeviceToDevice, c->stream));
    }
    st.finish();
    return 0;
  });
}

int bayens_loglik_rows_basis(bayens_ctx* c, int loc, int B, int N, int R_obs, const double* ll_basis,
                             const double* logdet, const double* obs, double* ll_mean) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R_obs <= 0 || !ll_basis || !logdet || !obs || !ll_mean)
      return fail(c, BAYENS_E_ARG, "loglik_rows_basis: bad argument");
    Stage st(c, loc);
    const double* db = st.in(ll_basis, (size_t)B * 2 * N);
    const double* dld = st.in(logdet, (size_t)B);
    const double* dobs = st.in(obs, (size_t)R_obs * N);
    double* dll = st.out(ll_mean, (size_t)B * N);
    c->prims->loglik_rows_basis(B, N, R_obs, db, dld, dobs, dll, c->small_scratch((size_t)B * 3));
    st.finish();
    return 0;
  });
}

int bayens_loglik_rows_basis_grouped(bayens_ctx* c, int loc, int B, int N, int R_obs, int n_groups, const double* ll_basis,
                                     const double* logdet, const double* obs, double* ll_mean) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R_obs <= 0 || n_groups <= 0 || B % n_groups || !ll_basis || !logdet || !obs || !ll_mean)
      return fail(c, BAYENS_E_ARG, "loglik_rows_basis_grouped: bad argument (B must be a multiple of n_groups)");
    Stage st(c, loc);
    const double* db = st.in(ll_basis, (size_t)B * 2 * N);
    const double* dld = st.in(logdet, (size_t)B);
    const double* dobs = st.in(obs, (size_t)n_groups * R_obs * N);
    double* dll = st.out(ll_mean, (size_t)B * N);
    c->prims->loglik_rows_basis(B, N, R_obs, db, dld, dobs, dll, c->small_scratch((size_t)B * 3), n_groups);
    st.finish();
    return 0;
  });
}

int bayens_normal_loglik_rows(bayens_ctx* c, int loc, int B, int N, int R_obs, const double* mu,
                              const double* scale, const double* obs, double* ll_mean) {
  return guarded(c, [&]() {
    if (B <= 0 || N <= 0 || R_obs <= 0 || !mu || !scale || !obs || !ll_mean)
      return fail(c, BAYENS_E_ARG, "normal_loglik_rows: bad argument");
    Stage st(c, loc);
    const double* dmu = st.in(mu, (size_t)B * N);
    const double* dsc = st.in(scale, (size_t)B * N);
    const double* dobs = st.in(obs, (size_t)R_obs * N);
    double* dll = st.out(ll_mean, (size_t)B * N);
    c->prims->normal_loglik_rows(B, N, R_obs, dmu, dsc, dobs, dll);
    st.finish();
    return 0;
  });
}

// LogLikelihoodWeight + Barycentre in one call (the reference's weighted posterior: weights.py:87-131 followed by
// ensemble_scheme.py:43-81): log-likelihood rows from the basis, the members' marginal variances from their
// covariances, normalisation over the members and the barycentre fixed point are queued back to back and the host
// waits once - the four separate entry points synchronise after each step.
int bayens_weighted_posterior(bayens_ctx* c, int loc, int M, int N, int R_obs, const double* ll_basis,
                              const double* logdet, const double* obs, const double* mu, const double* cov,
                              double c_std, double tol, double init_var, double* w, double* mu_out, double* var_out,
                              int* n_not_converged) {
  return guarded(c, [&]() {
    if (M <= 0 || N <= 0 || R_obs <= 0 || !ll_basis || !logdet || !obs || !mu || !cov || !mu_out || !var_out)
      return fail(c, BAYENS_E_ARG, "weighted_posterior: bad argument");
    Stage st(c, loc);
    const double* db = st.in(ll_basis, (size_t)M * 2 * N);
    const double* dld = st.in(logdet, (size_t)M);
    const double* dobs = st.in(obs, (size_t)R_obs * N);
    const double* dmu = st.in(mu, (size_t)M * N);
    const double* dcov = st.in(cov, (size_t)M * N * N);
    double* dw = w ? st.out(w, (size_t)M * N) : nullptr;
    double* dbm = st.out(mu_out, (size_t)N);
    double* dbv = st.out(var_out, (size_t)N);
    // persistent scratch: [M*3] dots | [1] not-converged count (as int) | [M*N] ll rows | [M*N] variances | [M*N] w
    double* sc = c->small_scratch((size_t)M * 3 + 2 + 3 * (size_t)M * N);
    double* dots = sc;
    int* dbad = reinterpret_cast<int*>(sc + (size_t)M * 3);
    double* dll = sc + (size_t)M * 3 + 2;
    double* dvar = dll + (size_t)M * N;
    if (!dw) dw = dvar + (size_t)M * N;
    c->prims->loglik_rows_basis(M, N, R_obs, db, dld, dobs, dll, dots);
    c->prims->extract_diag(M, N, dcov, dvar);
    c->prims->weights_normalise(M, N, dll, c_std, dw, nullptr);
    c->prims->barycentre(M, N, dw, dmu, dvar, tol, init_var, dbm, dbv, dbad);
    int hbad = 0;
    BAYENS_HIP(hipMemcpyAsync(&hbad, dbad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    if (n_not_converged) *n_not_converged = hbad;
    return 0;
  });
}

int bayens_weights_normalise(bayens_ctx* c, int loc, int M, int P, const double* ll_mean, double c_std, int scheme,
                             double* w, double* w_raw) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || !ll_mean || !w) return fail(c, BAYENS_E_ARG, "weights_normalise: bad argument");
    if (scheme != 0) return fail(c, BAYENS_E_ARG, "weights_normalise: only scheme 0 (exp) runs on the device");
    Stage st(c, loc);
    const double* dll = st.in(ll_mean, (size_t)M * P);
    double* dw = st.out(w, (size_t)M * P);
    double* draw = st.out(w_raw, (size_t)M * P);
    c->prims->weights_normalise(M, P, dll, c_std, dw, draw);
    st.finish();
    return 0;
  });
}

int bayens_w2_temporal(bayens_ctx* c, int loc, int M, int P, const double* mean, const double* var, double* w) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || !mean || !var || !w) return fail(c, BAYENS_E_ARG, "w2_temporal: bad argument");
    Stage st(c, loc);
    const double* dm = st.in(mean, (size_t)M * P);
    const double* dv = st.in(var, (size_t)M * P);
    double* dw = st.out(w, (size_t)M * P);
    c->prims->w2_temporal(M, P, dm, dv, dw);
    st.finish();
    return 0;
  });
}

int bayens_w2_spatial(bayens_ctx* c, int loc, int M, int T, int G, const double* mean, const double* scale, double* w) {
  return guarded(c, [&]() {
    if (M <= 0 || T <= 0 || G <= 0 || !mean || !scale || !w) return fail(c, BAYENS_E_ARG, "w2_spatial: bad argument");
    Stage st(c, loc);
    const double* dm = st.in(mean, (size_t)M * T * G);
    const double* dv = st.in(scale, (size_t)M * T * G);
    double* dw = st.out(w, (size_t)M * G);
    c->prims->w2_spatial(M, T, G, dm, dv, dw);
    st.finish();
    return 0;
  });
}

namespace {
// All-pairs Gaussian W2 table and weights.  sharded: every rank holds ALL M members (the caller all-gathered the
// factors); the M(M-1)/2 polar iterations are dealt over the ranks of the communicator by anchor member i (i mod
// nranks: the batches of an anchor have M-1-i pairs, so interleaving balances the work), each rank fills its part of
// the nuclear-norm table and one ncclAllReduce (sum, M*M doubles) completes it everywhere.
// v_b <- v_b / ||v_b||, norm[b] = ||v_b|| (one workgroup per member)
__global__ __launch_bounds__(256) void vec_normalise_kernel(int n, double* v, double* norm) {
  __shared__ double red[256];
  double* x = v + (size_t)blockIdx.x * n;
  double s = 0.0;
  for (int t = threadIdx.x; t < n; t += 256) s += x[t] * x[t];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int h = 128; h > 0; h >>= 1) {
    if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
    __syncthreads();
  }
  const double nrm = sqrt(red[0]);
  const double r = nrm > 0.0 ? 1.0 / nrm : 0.0;
  for (int t = threadIdx.x; t < n; t += 256) x[t] *= r;
  if (threadIdx.x == 0) norm[blockIdx.x] = nrm;
}

// lam[b] <- ||A^T A v|| after `steps` normalised power steps from v (A lower triangular, zeros above): an estimate FROM
// BELOW of the largest eigenvalue of A_b^T A_b.  v [nb][n] is overwritten, y [nb][n] scratch, ones [nb][n] = 1.
void tri_power(HipPrims& P, hipStream_t stream, int nb, int n, Mat A, double* v, double* y, const double* ones, double* lam,
               int steps) {
  for (int it = 0; it < steps; ++it) {
    P.resid(nb, n, A, v, nullptr, y);                 // y = A v
    P.lt_dvec(nb, n, A, y, ones, 1.0, 0.0, v);        // v = A^T y
    hipLaunchKernelGGL(vec_normalise_kernel, dim3(nb), dim3(256), 0, stream, n, v, lam);
    BAYENS_HIP(hipGetLastError());
  }
}

// packed[p] (pairs i < j in row-major order) -> table[i][j]
__global__ void scatter_pairs_kernel(int M, int n_pairs, const double* packed, double* table) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  int i = 0, first = 0;                     // first = index of pair (i, i + 1)
  while (first + (M - 1 - i) <= p) { first += M - 1 - i; ++i; }
  table[(size_t)i * M + (i + 1 + p - first)] = packed[p];
}

int w2_single_impl(bayens_ctx* c, int loc, int M, int N, const double* mu, const double* chol, const double* sd,
                   double* w, double* D, bool sharded) {
  return guarded(c, [&]() {
    if (M <= 0 || N <= 0 || !mu || !w || (!chol && !sd)) return fail(c, BAYENS_E_ARG, "w2_single: bad argument");
    if (sharded && (!c->comm || !c->rccl.AllReduce))
      return fail(c, BAYENS_E_COMM, "w2_single_sharded: communicator not initialised");
    const int nranks = sharded ? c->nranks : 1, rank = sharded ? c->rank : 0;
    Stage st(c, loc);
    const double* dmu = st.in(mu, (size_t)M * N);
    double* dw = st.out(w, (size_t)M);
    double* dD = st.out(D, (size_t)M * M);
    double* gap = st.scratch<double>((size_t)M * M);
    HipPrims& P = *c->prims;
    if (!chol) {
      const double* dsd = st.in(sd, (size_t)M * N);
      double* dt = st.scratch<double>((size_t)M * M);
      P.pair_terms(M, N, dmu, dsd, gap, dt);
      P.w2_single_finish(M, nullptr, nullptr, gap, dt, dD, dw);
      st.finish();
      return 0;
    }
    const double* dch = st.in(chol, (size_t)M * N * N);
    const int Np = pad_to_leaf(N);
    ensure_ws(c, M, Np);
    FitWorkspace ws = ws_view(c, 0, M);
    BAYENS_HIP(hipMemsetAsync(ws.info, 0, (size_t)M * sizeof(int), c->stream));
    double* Lpad = st.scratch<double>((size_t)M * Np * Np);
    Mat Lp{Lpad, Np, (long)Np * Np};
    P.pad_in(M, N, Np, dch, Lp);                       // identity padding: contributes exactly Np - N to every norm
    double* tr = st.scratch<double>((size_t)M);
    double* nuc = st.scratch<double>((size_t)M * M + 1);
    double* rel = st.scratch<double>((size_t)M);
    P.dot_all(M, Np, Lp, Lp, tr);                      // tr S_i = ||L_i||_F^2 (+ padding)
    P.pair_terms(M, N, dmu, nullptr, gap, nullptr);
    // The interval of every factor's singular values, once per member (a few fit iterations' worth of work for all M):
    //   rigorous:  hi_m = sqrt(||L||_1 ||L||_inf) >= sigma_max(L_m),   lo_m = 1 / sqrt(||L^-1||_1 ||L^-1||_inf) <= sigma_min(L_m);
    //   estimated: smax_m, smin_m from W2_POWER_STEPS power steps on L^T L and L^-T L^-1 (the explicit inverse is there
    //              anyway), which approach the true values from inside the interval.
    // A pair's product L_j^T L_i has its singular values in [smin_i smin_j, smax_i smax_j]: the polar iteration starts from
    // X_0 = M / min(||M||_F, 1.21 smax_i smax_j, hi_i hi_j) and from the lower bound ell0 = smin_i smin_j / (4 x that
    // scale) instead of a blanket 1e-7 - on posterior covariances that is one of its five steps.  The estimates only
    // steer the weights; what ends the iteration is the size of its steps (polar_nuclear), and an ell0 that is too
    // optimistic delays singular values whose share of the nuclear norm is below ell0 itself.
    constexpr int W2_POWER_STEPS = 12;
    std::vector<double> hi(M, 0.0), lo(M, 0.0), smax(M, 0.0), smin(M, 0.0);
    {
      double* bounds = st.scratch<double>((size_t)M * 6);
      double* vec = st.scratch<double>((size_t)M * Np * 3);
      double *v = vec, *y = vec + (size_t)M * Np, *ones = vec + (size_t)M * Np * 2;
      std::vector<double> h0((size_t)M * Np), h1((size_t)M * Np, 1.0);
      unsigned long long z = 0x9E3779B97F4A7C15ull;
      for (auto& x : h0) {                             // fixed pseudo-random start: every component of both signs
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        x = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 0.75;
      }
      BAYENS_HIP(hipMemsetAsync(bounds, 0, (size_t)M * 6 * sizeof(double), c->stream));
      BAYENS_HIP(hipMemcpyAsync(ones, h1.data(), h1.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
      BAYENS_HIP(hipMemcpyAsync(v, h0.data(), h0.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
      P.tri_absnorms(M, Np, Lp, bounds);
      tri_power(P, c->stream, M, Np, Lp, v, y, ones, bounds + (size_t)M * 4, W2_POWER_STEPS);
      BAYENS_HIP(hipMemsetAsync(ws.info, 0, (size_t)M * sizeof(int), c->stream));
      P.tri_inv64(M, Np, Lp, ws.invL, ws.inv_stride(), ws.info);
      P.identity(M, Np, ws.T2);
      trsm_rln(P, M, Np, Np, Lp, ws.invL, ws.inv_stride(), ws.T2, true);
      P.tri_absnorms(M, Np, ws.T2, bounds + (size_t)M * 2);
      BAYENS_HIP(hipMemcpyAsync(v, h0.data(), h0.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
      tri_power(P, c->stream, M, Np, ws.T2, v, y, ones, bounds + (size_t)M * 5, W2_POWER_STEPS);
      std::vector<double> hb((size_t)M * 6);
      BAYENS_HIP(hipMemcpyAsync(hb.data(), bounds, hb.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
      std::vector<int> hinfo(M);
      BAYENS_HIP(hipMemcpyAsync(hinfo.data(), ws.info, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      BAYENS_HIP(hipStreamSynchronize(c->stream));
      for (int m = 0; m < M; ++m)
        if (hinfo[m])
          return fail(c, BAYENS_E_NOTPD, "w2_single: the factor of member " + std::to_string(m) +
                                             " has a diagonal entry that is not positive and finite (row " +
                                             std::to_string(hinfo[m]) + ")");
      for (int m = 0; m < M; ++m) {
        hi[m] = std::sqrt(hb[(size_t)m * 2] * hb[(size_t)m * 2 + 1]);
        lo[m] = 1.0 / std::sqrt(hb[(size_t)(M + m) * 2] * hb[(size_t)(M + m) * 2 + 1]);
        smax[m] = std::fmin(hi[m], 1.1 * std::sqrt(hb[(size_t)M * 4 + m]));
        smin[m] = std::fmax(lo[m], 1.0 / std::sqrt(hb[(size_t)M * 5 + m]));
      }
    }
    const bool trace = getenv("BAYENS_W2_TRACE") != nullptr;
    // nuc[M*M] is followed by one more double: the number of batches whose polar iteration did not converge (summed
    // over the ranks by the same all-reduce, so that every rank takes the same decision)
    BAYENS_HIP(hipMemsetAsync(nuc, 0, ((size_t)M * M + 1) * sizeof(double), c->stream));
    // The M(M-1)/2 pairs in row-major order (i < j), M slots at a time: every batch of the iteration is full (the last
    // one excepted) whichever anchors its pairs belong to - only the products L_j^T L_i that fill the slots are issued
    // per anchor.  Batches are dealt over the ranks.
    const int n_pairs = M * (M - 1) / 2;
    std::vector<int> pi((size_t)n_pairs), pj((size_t)n_pairs);
    std::vector<double> hcap((size_t)n_pairs > 0 ? n_pairs : 1);
    for (int i = 0, p = 0; i + 1 < M; ++i)
      for (int j = i + 1; j < M; ++j, ++p) { pi[p] = i; pj[p] = j; hcap[p] = smax[i] * smax[j]; }
    double* packed = st.scratch<double>((size_t)n_pairs + 1);
    double* dcap = st.scratch<double>((size_t)n_pairs + 1);
    BAYENS_HIP(hipMemsetAsync(packed, 0, ((size_t)n_pairs + 1) * sizeof(double), c->stream));
    BAYENS_HIP(hipMemcpyAsync(dcap, hcap.data(), (size_t)n_pairs * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    double not_converged = 0.0;
    for (int p0 = 0, chunk = 0; p0 < n_pairs; p0 += M, ++chunk) {
      if (chunk % nranks != rank) continue;
      const int nb = std::min(M, n_pairs - p0);
      FitWorkspace wsi = ws;
      wsi.nb = nb;
      double ell0 = 1.0;
      for (int s0 = 0; s0 < nb;) {                     // one product launch per run of slots that share the anchor
        const int i = pi[p0 + s0];
        int len = 1;
        while (s0 + len < nb && pi[p0 + s0 + len] == i) ++len;
        const int j0 = pj[p0 + s0];
        P.gemm(len, Np, Np, Np, 1.0, Mat{Lpad + (size_t)j0 * Np * Np, Np, (long)Np * Np}, Mat{Lpad + (size_t)i * Np * Np, Np, 0},
               0.0, Mat{ws.W0.p + (size_t)s0 * ws.W0.stride, ws.W0.ld, ws.W0.stride}, GEMM_TA | GEMM_KTRIM_TN_LOWER, 0.0);
        for (int q = 0; q < len; ++q) ell0 = std::fmin(ell0, 0.25 * (smin[i] * smin[j0 + q]) / (smax[i] * smax[j0 + q]));
        s0 += len;
      }
      const bool bracketed = ell0 > 1e-7 && ell0 < 1.0;      // (NaN / non-finite brackets fall back to the blanket bound)
      if (!bracketed) ell0 = 1e-7;
      const int its = polar_nuclear(P, wsi, nb, packed + p0, rel, ell0, 1e-7, 16, bracketed ? dcap + p0 : nullptr, 1.0,
                                    bracketed ? 1e-12 : 0.0);
      if (trace) fprintf(stderr, "w2_single: pairs %d..%d, ell0 %.3g, %d steps\n", p0, p0 + nb - 1, ell0, its);
      if (its <= 0) not_converged += 1.0;
    }
    if (n_pairs > 0) {
      hipLaunchKernelGGL(scatter_pairs_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, c->stream, M, n_pairs, packed, nuc);
      BAYENS_HIP(hipGetLastError());
    }
    if (sharded) {
      BAYENS_HIP(hipMemcpyAsync(nuc + (size_t)M * M, &not_converged, sizeof(double), hipMemcpyHostToDevice, c->stream));
      BAYENS_HIP(hipStreamSynchronize(c->stream));     // (the source is a stack variable that is overwritten below)
      ncclResult_t r = c->rccl.AllReduce(nuc, nuc, (size_t)M * M + 1, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream);
      if (r != 0) return fail(c, BAYENS_E_COMM, "ncclAllReduce failed");
      BAYENS_HIP(hipMemcpyAsync(&not_converged, nuc + (size_t)M * M, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    // remove the identity padding's share (Np - N) from traces and nuclear norms: both enter as tr_i + tr_j - 2 nuc,
    // i.e. 2 (Np - N) - 2 (Np - N) = 0, so no correction is needed in D.
    P.w2_single_finish(M, tr, nuc, gap, nullptr, dD, dw);
    std::vector<int> hinfo(M, 0);
    BAYENS_HIP(hipMemcpyAsync(hinfo.data(), ws.info, (size_t)M * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    for (int b = 0; b < M; ++b)
      if (hinfo[b]) return fail(c, BAYENS_E_NOTPD, "w2_single: a polar-iteration Cholesky failed");
    if (not_converged > 0.0)
      return fail(c, BAYENS_E_NOCONV, "w2_single: the polar iteration of at least one pair did not converge (ill-conditioned "
                                      "or non-finite covariance factors); w and D are not valid");
    return 0;
  });
}
}  // namespace

int bayens_w2_single(bayens_ctx* c, int loc, int M, int N, const double* mu, const double* chol, const double* sd,
                     double* w, double* D) {
  return w2_single_impl(c, loc, M, N, mu, chol, sd, w, D, false);
}

int bayens_w2_single_sharded(bayens_ctx* c, int loc, int M, int N, const double* mu, const double* chol, double* w,
                             double* D) {
  if (!chol) return c ? fail(c, BAYENS_E_ARG, "w2_single_sharded: needs the Cholesky factors") : BAYENS_E_ARG;
  return w2_single_impl(c, loc, M, N, mu, chol, nullptr, w, D, true);
}

int bayens_barycentre(bayens_ctx* c, int loc, int M, int P, const double* w, const double* mean, const double* var,
                      double tol, double init_var, double* mu_out, double* var_out, int* n_not_converged) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || !w || !mean || !var || !mu_out || !var_out)
      return fail(c, BAYENS_E_ARG, "barycentre: bad argument");
    Stage st(c, loc);
    const double* dw = st.in(w, (size_t)M * P);
    const double* dm = st.in(mean, (size_t)M * P);
    const double* dv = st.in(var, (size_t)M * P);
    double* dmu = st.out(mu_out, (size_t)P);
    double* dvar = st.out(var_out, (size_t)P);
    int* dbad = reinterpret_cast<int*>(c->small_scratch(1));     // persistent: no allocation on the latency path
    c->prims->barycentre(M, P, dw, dm, dv, tol, init_var, dmu, dvar, dbad);
    int hbad = 0;
    BAYENS_HIP(hipMemcpyAsync(&hbad, dbad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    if (n_not_converged) *n_not_converged = hbad;
    return 0;
  });
}

int bayens_weighted_model_mean(bayens_ctx* c, int loc, int M, int R_max, const int* R, int P, const double* data,
                               const double* w, double* mu_out, double* var_out) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || R_max <= 0 || !R || !data || !w || !mu_out || !var_out)
      return fail(c, BAYENS_E_ARG, "weighted_model_mean: bad argument");
    Stage st(c, loc);
    const int* dR = st.in(R, (size_t)M);
    const double* dd = st.in(data, (size_t)M * R_max * P);
    const double* dw = st.in(w, (size_t)M * P);
    double* dmu = st.out(mu_out, (size_t)P);
    double* dvar = st.out(var_out, (size_t)P);
    c->prims->weighted_model_mean(M, R_max, dR, P, dd, dw, dmu, dvar);
    st.finish();
    return 0;
  });
}

int bayens_score_weights(bayens_ctx* c, int loc, int kind, int M, int P, int R_obs, const double* mean,
                         const double* var, const double* obs, double* w, double* score) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || R_obs <= 0 || kind < 0 || kind > 1 || !mean || !var || !obs || !w)
      return fail(c, BAYENS_E_ARG, "score_weights: bad argument");
    Stage st(c, loc);
    const double* dm = st.in(mean, (size_t)M * P);
    const double* dv = st.in(var, (size_t)M * P);
    const double* dobs = st.in(obs, (size_t)R_obs * P);
    double* dw = st.out(w, (size_t)M * P);
    double* ds = st.out(score, (size_t)M * P);
    c->prims->score_weights(kind, M, P, R_obs, dm, dv, dobs, dw, ds);
    st.finish();
    return 0;
  });
}

int bayens_inverse_square_weights(bayens_ctx* c, int loc, int M, int R_max, const int* R, int P, const double* data,
                                  int R_obs, const double* obs, double* w) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || R_max <= 0 || R_obs <= 0 || !R || !data || !obs || !w)
      return fail(c, BAYENS_E_ARG, "inverse_square_weights: bad argument");
    Stage st(c, loc);
    const int* dR = st.in(R, (size_t)M);
    const double* dd = st.in(data, (size_t)M * R_max * P);
    const double* dobs = st.in(obs, (size_t)R_obs * P);
    double* dw = st.out(w, (size_t)M * P);
    c->prims->inverse_square_weights(M, R_max, dR, P, dd, R_obs, dobs, dw);
    st.finish();
    return 0;
  });
}

int bayens_multi_model_mean(bayens_ctx* c, int loc, int M, int R_max, const int* R, int P, const double* data,
                            double* mu_out, double* var_out) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || R_max <= 0 || !R || !data || !mu_out || !var_out)
      return fail(c, BAYENS_E_ARG, "multi_model_mean: bad argument");
    Stage st(c, loc);
    const int* dR = st.in(R, (size_t)M);
    const double* dd = st.in(data, (size_t)M * R_max * P);
    double* dmu = st.out(mu_out, (size_t)P);
    double* dvar = st.out(var_out, (size_t)P);
    c->prims->multi_model_mean(M, R_max, dR, P, dd, dmu, dvar);
    st.finish();
    return 0;
  });
}

int bayens_dba_batch(bayens_ctx* c, int loc, int B, int R, int N, const double* series, int n_iterations, double* center,
                     int* medoid) {
  return guarded(c, [&]() {
    if (B <= 0 || R <= 0 || N <= 0 || n_iterations < 0 || !series || !center) return fail(c, BAYENS_E_ARG, "dba_batch: bad argument");
    Stage st(c, loc);
    const double* ds = st.in(series, (size_t)B * R * N);
    double* dc = st.out(center, (size_t)B * N);
    int* dm = medoid ? st.out(medoid, (size_t)B) : st.scratch<int>((size_t)B);
    double* sq = st.scratch<double>((size_t)B * R * R);
    signed char* path = st.scratch<signed char>((size_t)B * R * N * N);
    int* jlo = st.scratch<int>((size_t)B * R * N);
    int* jhi = st.scratch<int>((size_t)B * R * N);
    c->prims->dba_batch(B, R, N, ds, n_iterations, dc, dm, sq, path, jlo, jhi);
    st.finish();
    return 0;
  });
}

// ---- RCCL ---------------------------------------------------------------------------------------
int bayens_comm_available(bayens_ctx* c) {
  return guarded(c, [&]() {
    std::string err;
    if (!c->rccl.load(err)) return fail(c, BAYENS_E_COMM, err);
    return 0;
  });
}

int bayens_comm_unique_id(bayens_ctx* c, void* id128) {
  return guarded(c, [&]() {
    if (!id128) return fail(c, BAYENS_E_ARG, "comm_unique_id: null buffer");
    std::string err;
    if (!c->rccl.load(err)) return fail(c, BAYENS_E_COMM, err);
    ncclUniqueId id;
    ncclResult_t r = c->rccl.GetUniqueId(&id);
    if (r != 0) return fail(c, BAYENS_E_COMM, "ncclGetUniqueId failed");
    memcpy(id128, &id, sizeof(id));
    return 0;
  });
}

int bayens_comm_init(bayens_ctx* c, int rank, int nranks, const void* id128) {
  return guarded(c, [&]() {
    if (!id128 || nranks <= 0 || rank < 0 || rank >= nranks) return fail(c, BAYENS_E_ARG, "comm_init: bad argument");
    std::string err;
    if (!c->rccl.load(err)) return fail(c, BAYENS_E_COMM, err);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = c->rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != 0) {
      c->comm = nullptr;
      return fail(c, BAYENS_E_COMM, std::string("ncclCommInitRank: ") +
                                        (c->rccl.GetErrorString ? c->rccl.GetErrorString(r) : "error"));
    }
    c->rank = rank; c->nranks = nranks;
    return 0;
  });
}

int bayens_comm_destroy(bayens_ctx* c) {
  return guarded(c, [&]() {
    if (c->comm) { c->rccl.CommDestroy(c->comm); c->comm = nullptr; }
    return 0;
  });
}

int bayens_allgather_rows(bayens_ctx* c, const double* local_dev, double* all_dev, int M_loc, int P) {
  return guarded(c, [&]() {
    if (!c->comm) return fail(c, BAYENS_E_COMM, "allgather_rows: communicator not initialised");
    if (!local_dev || !all_dev || M_loc <= 0 || P <= 0) return fail(c, BAYENS_E_ARG, "allgather_rows: bad argument");
    ncclResult_t r = c->rccl.AllGather(local_dev, all_dev, (size_t)M_loc * P, NCCL_FLOAT64, c->comm, c->stream);
    if (r != 0) return fail(c, BAYENS_E_COMM, "ncclAllGather failed");
    return 0;
  });
}

int bayens_comm_info(bayens_ctx* c, int* rank, int* nranks, char* lib_path, int lib_path_len) {
  return guarded(c, [&]() {
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->comm ? c->nranks : 0;
    if (lib_path && lib_path_len > 0) snprintf(lib_path, lib_path_len, "%s", c->rccl.path.c_str());
    return 0;
  });
}

// The member exchange of the sharded weighted posterior.  Members are dealt to the ranks in contiguous blocks, the
// first M % nranks ranks holding one more (bayens_shard_bounds); every rank contributes rows = ceil(M / nranks)
// packed [3][P] rows (log-likelihood, mean, variance; unused rows are never read) to ONE ncclAllGather and compacts
// the received [nranks][rows][3][P] image into the three [M][P] tables.
}  // extern "C"
namespace {
__global__ void pack_rows_kernel(int M_loc, int P, const double* ll, const double* mean, const double* var, double* packed) {
  const int m = blockIdx.y, which = blockIdx.z, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P || m >= M_loc) return;
  const double* src = which == 0 ? ll : (which == 1 ? mean : var);
  packed[((size_t)m * 3 + which) * P + t] = src[(size_t)m * P + t];
}
__global__ void unpack_rows_kernel(int M, int nranks, int rows, int P, const double* gathered, double* ll, double* mean,
                                   double* var) {
  const int m = blockIdx.y, which = blockIdx.z, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P || m >= M) return;
  const int base = M / nranks, extra = M % nranks;          // ranks < extra own base + 1 members
  int r, loc;
  if (m < (base + 1) * extra) { r = m / (base + 1); loc = m - r * (base + 1); }
  else { const int q = m - (base + 1) * extra; r = extra + q / base; loc = q - (r - extra) * base; }
  double* dst = which == 0 ? ll : (which == 1 ? mean : var);
  dst[(size_t)m * P + t] = gathered[(((size_t)r * rows + loc) * 3 + which) * P + t];
}
}  // namespace
extern "C" {

int bayens_shard_bounds(int M, int rank, int nranks, int* lo, int* hi) {
  if (M < 0 || nranks <= 0 || rank < 0 || rank >= nranks || !lo || !hi) return BAYENS_E_ARG;
  const int base = M / nranks, extra = M % nranks;
  *lo = rank * base + (rank < extra ? rank : extra);
  *hi = *lo + base + (rank < extra ? 1 : 0);
  return 0;
}

int bayens_allgather_member_rows(bayens_ctx* c, int M, int P, int M_loc, const double* ll_loc, const double* mean_loc,
                                 const double* var_loc, double* ll_all, double* mean_all, double* var_all) {
  return guarded(c, [&]() {
    if (!c->comm) return fail(c, BAYENS_E_COMM, "allgather_member_rows: communicator not initialised");
    if (M <= 0 || P <= 0 || !ll_loc || !mean_loc || !var_loc || !ll_all || !mean_all || !var_all)
      return fail(c, BAYENS_E_ARG, "allgather_member_rows: bad argument");
    int lo = 0, hi = 0;
    bayens_shard_bounds(M, c->rank, c->nranks, &lo, &hi);
    if (M_loc != hi - lo)
      return fail(c, BAYENS_E_ARG, "allgather_member_rows: M_loc is not this rank's block of the M members (bayens_shard_bounds)");
    const int rows = (M + c->nranks - 1) / c->nranks;
    const size_t per_rank = (size_t)rows * 3 * P;
    double* send = c->exchange_scratch(per_rank * (c->nranks + 1));
    double* recv = send + per_rank;
    const dim3 block(256);
    if (M_loc > 0) {
      hipLaunchKernelGGL(pack_rows_kernel, dim3((P + 255) / 256, M_loc, 3), block, 0, c->stream, M_loc, P, ll_loc, mean_loc,
                         var_loc, send);
      BAYENS_HIP(hipGetLastError());
    }
    ncclResult_t r = c->rccl.AllGather(send, recv, per_rank, NCCL_FLOAT64, c->comm, c->stream);
    if (r != 0) return fail(c, BAYENS_E_COMM, "ncclAllGather failed");
    hipLaunchKernelGGL(unpack_rows_kernel, dim3((P + 255) / 256, M, 3), block, 0, c->stream, M, c->nranks, rows, P, recv,
                       ll_all, mean_all, var_all);
    BAYENS_HIP(hipGetLastError());
    return 0;
  });
}

// include/bayens_diag.h: the exchange's index arithmetic (pack per fictitious rank -> gathered image -> unpack), no RCCL
int bayens_test_member_rows_exchange(bayens_ctx* c, int M, int P, int nranks, const double* ll, const double* mean,
                                     const double* var, double* ll_out, double* mean_out, double* var_out) {
  return guarded(c, [&]() {
    if (M <= 0 || P <= 0 || nranks <= 0 || !ll || !mean || !var || !ll_out || !mean_out || !var_out)
      return fail(c, BAYENS_E_ARG, "test_member_rows_exchange: bad argument");
    const int rows = (M + nranks - 1) / nranks;
    const size_t per_rank = (size_t)rows * 3 * P, tab = (size_t)M * P;
    double* d = nullptr;
    BAYENS_HIP(hipMalloc((void**)&d, (6 * tab + per_rank * nranks) * sizeof(double)));
    double *din = d, *dout = d + 3 * tab, *gathered = d + 6 * tab;
    int rc = 0;
    auto body = [&]() -> int {
      BAYENS_HIP(hipMemcpyAsync(din, ll, tab * 8, hipMemcpyHostToDevice, c->stream));
      BAYENS_HIP(hipMemcpyAsync(din + tab, mean, tab * 8, hipMemcpyHostToDevice, c->stream));
      BAYENS_HIP(hipMemcpyAsync(din + 2 * tab, var, tab * 8, hipMemcpyHostToDevice, c->stream));
      BAYENS_HIP(hipMemsetAsync(gathered, 0xFF, per_rank * nranks * 8, c->stream));       // all-ones words = NaN
      BAYENS_HIP(hipMemsetAsync(dout, 0xFF, 3 * tab * 8, c->stream));
      const dim3 block(256);
      for (int r = 0; r < nranks; ++r) {
        int lo = 0, hi = 0;
        bayens_shard_bounds(M, r, nranks, &lo, &hi);
        if (hi > lo)
          hipLaunchKernelGGL(pack_rows_kernel, dim3((P + 255) / 256, hi - lo, 3), block, 0, c->stream, hi - lo, P,
                             din + (size_t)lo * P, din + tab + (size_t)lo * P, din + 2 * tab + (size_t)lo * P,
                             gathered + per_rank * r);
      }
      hipLaunchKernelGGL(unpack_rows_kernel, dim3((P + 255) / 256, M, 3), block, 0, c->stream, M, nranks, rows, P, gathered,
                         dout, dout + tab, dout + 2 * tab);
      BAYENS_HIP(hipGetLastError());
      BAYENS_HIP(hipMemcpyAsync(ll_out, dout, tab * 8, hipMemcpyDeviceToHost, c->stream));
      BAYENS_HIP(hipMemcpyAsync(mean_out, dout + tab, tab * 8, hipMemcpyDeviceToHost, c->stream));
      BAYENS_HIP(hipMemcpyAsync(var_out, dout + 2 * tab, tab * 8, hipMemcpyDeviceToHost, c->stream));
      BAYENS_HIP(hipStreamSynchronize(c->stream));
      return 0;
    };
    try { rc = body(); } catch (...) { (void)hipFree(d); throw; }
    (void)hipFree(d);
    return rc;
  });
}

int bayens_comm_max(bayens_ctx* c, double* value) {
  return guarded(c, [&]() {
    if (!c->comm || !c->rccl.AllReduce) return fail(c, BAYENS_E_COMM, "comm_max: communicator not initialised");
    if (!value) return fail(c, BAYENS_E_ARG, "comm_max: null value");
    double* d = c->small_scratch(2) + 1;      // (slot 0 is the barycentre's not-converged counter)
    BAYENS_HIP(hipMemcpyAsync(d, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    ncclResult_t r = c->rccl.AllReduce(d, d, 1, NCCL_FLOAT64, NCCL_MAX, c->comm, c->stream);
    if (r != 0) return fail(c, BAYENS_E_COMM, "ncclAllReduce failed");
    BAYENS_HIP(hipMemcpyAsync(value, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    return 0;
  });
}

int bayens_bench_diag16(bayens_ctx* c, int variant, int waves, int reps, double* cycles) {
  return guarded(c, [&]() {
    if (waves < 1 || waves > 16 || reps < 1 || !cycles) return fail(c, BAYENS_E_ARG, "bench_diag16: bad argument");
    double* d = c->small_scratch(4);
    smalln::diag_bench(c->stream, variant, waves, reps, d);
    BAYENS_HIP(hipMemcpyAsync(cycles, d, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    return 0;
  });
}

int bayens_small_stamps(bayens_ctx* c, int enable, unsigned long long* stamps32) {
  return guarded(c, [&]() {
    if (enable && !c->small_stamps) {
      BAYENS_HIP(hipMalloc((void**)&c->small_stamps, 160 * sizeof(unsigned long long)));
      BAYENS_HIP(hipMemset(c->small_stamps, 0, 160 * sizeof(unsigned long long)));
    }
    if (stamps32 && c->small_stamps) {
      BAYENS_HIP(hipStreamSynchronize(c->stream));
      BAYENS_HIP(hipMemcpy(stamps32, c->small_stamps, 32 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    if (!enable && c->small_stamps) { (void)hipFree(c->small_stamps); c->small_stamps = nullptr; }
    return 0;
  });
}

int bayens_potrf256_stamps(bayens_ctx* c, unsigned long long* stamps128) {
  return guarded(c, [&]() {
    if (!stamps128) return fail(c, BAYENS_E_ARG, "potrf256_stamps: null output");
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    potrf256n::potrf256_stamps(stamps128);
    return 0;
  });
}

int bayens_small_wave_stamps(bayens_ctx* c, unsigned long long* stamps128) {
  return guarded(c, [&]() {
    if (!stamps128 || !c->small_stamps) return fail(c, BAYENS_E_ARG, "small_wave_stamps: recording is off (bayens_small_stamps(ctx, 1, NULL) first)");
    BAYENS_HIP(hipStreamSynchronize(c->stream));
    BAYENS_HIP(hipMemcpy(stamps128, c->small_stamps + 32, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
  });
}

// ---- diagnostics for the GPU unit tests ----------------------------------------------------------
int bayens_test_gemm(bayens_ctx* c, int nb, int m, int n, int k, double alpha, const double* A, int lda,
                     const double* B, int ldb, double beta, double* C, int ldc, int flags, double diag_add) {
  return guarded(c, [&]() {
    const bool ta = flags & GEMM_TA, tb = flags & GEMM_TB;
    const size_t an = (size_t)(ta ? k : m) * lda, bn = (size_t)(tb ? n : k) * ldb, cn = (size_t)m * ldc;
    Stage st(c, BAYENS_HOST);
    const double* dA = st.in(A, an * nb);
    const double* dB = st.in(B, bn * nb);
    double* dC = st.scratch<double>(cn * nb);
    BAYENS_HIP(hipMemcpyAsync(dC, C, cn * nb * sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->prims->gemm(nb, m, n, k, alpha, Mat{const_cast<double*>(dA), lda, (long)an},
                   Mat{const_cast<double*>(dB), ldb, (long)bn}, beta, Mat{dC, ldc, (long)cn}, flags, diag_add);
    BAYENS_HIP(hipMemcpyAsync(C, dC, cn * nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    return 0;
  });
}

int bayens_bench_gemm(bayens_ctx* c, int nb, int m, int n, int k, int flags, int reps, double* ms_per_launch) {
  return guarded(c, [&]() {
    if (nb <= 0 || reps <= 0) return fail(c, BAYENS_E_ARG, "bench_gemm: bad argument");
    const bool ta = flags & GEMM_TA, tb = flags & GEMM_TB;
    const int lda = ta ? m : k, ldb = tb ? k : n;
    const size_t an = (size_t)(ta ? k : m) * lda, bn = (size_t)(tb ? n : k) * ldb, cn = (size_t)m * n;
    Stage st(c, BAYENS_DEVICE);
    double* dA = st.scratch<double>(an * nb);
    double* dB = st.scratch<double>(bn * nb);
    double* dC = st.scratch<double>(cn * nb);
    c->prims->fill_random(dA, an * nb, 1);
    c->prims->fill_random(dB, bn * nb, 2);
    c->prims->fill_random(dC, cn * nb, 3);
    Mat A{dA, lda, (long)an}, B{dB, ldb, (long)bn}, Cm{dC, n, (long)cn};
    double beta = 0.5;
    if (const char* env = getenv("BAYENS_BENCH_BETA")) beta = atof(env);
    c->prims->gemm(nb, m, n, k, 1.0, A, B, beta, Cm, flags, 0.0);     // warm-up
    BAYENS_HIP(hipEventRecord(c->t0, c->stream));
    for (int r = 0; r < reps; ++r) c->prims->gemm(nb, m, n, k, 1.0, A, B, beta, Cm, flags, 0.0);
    BAYENS_HIP(hipEventRecord(c->t1, c->stream));
    BAYENS_HIP(hipEventSynchronize(c->t1));
    float f = 0;
    BAYENS_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    if (ms_per_launch) *ms_per_launch = f / reps;
    return 0;
  });
}

int bayens_bench_leaf(bayens_ctx* c, int nb, int reps, double* us_per_launch, unsigned long long* stamps16) {
  return guarded(c, [&]() {
    Stage st(c, BAYENS_DEVICE);
    const int n = 64;
    double* dA = st.scratch<double>((size_t)nb * n * n);
    double* dS = st.scratch<double>((size_t)nb * n * n);
    double* inv = st.scratch<double>((size_t)nb * n * n);
    int* dinfo = st.scratch<int>(nb);
    unsigned long long* dst = st.scratch<unsigned long long>(16);
    BAYENS_HIP(hipMemsetAsync(dinfo, 0, nb * sizeof(int), c->stream));
    BAYENS_HIP(hipMemsetAsync(dst, 0, 16 * sizeof(unsigned long long), c->stream));
    c->prims->fill_random(dA, (size_t)nb * n * n, 7);
    Mat A{dA, n, (long)n * n}, S{dS, n, (long)n * n};
    // S = A A^T + 64 I  (SPD)
    c->prims->gemm(nb, n, n, n, 1.0, A, A, 0.0, S, GEMM_TB, 64.0);
    c->prims->copy(nb, n, S, A);
    // BAYENS_LEAF_SIDE=k: the stamped launch runs beside a trailing-update-sized product on another stream
    // (k = 1: 1536 x 1536 x 256 lower, the Cholesky's side-stream update; k = 2: a copy of the same size instead)
    const char* side_env = getenv("BAYENS_LEAF_SIDE");
    const int side = side_env ? atoi(side_env) : 0;
    double *sa = nullptr, *sc = nullptr;
    if (side) {
      const size_t an = (size_t)1536 * 256, cn = (size_t)1536 * 1536;
      sa = st.scratch<double>(an * 32);
      sc = st.scratch<double>(cn * 32 * 2);
      c->prims->fill_random(sa, an * 32, 5);
      c->prims->fill_random(sc, cn * 32 * 2, 6);
      BAYENS_HIP(hipStreamSynchronize(c->stream));
      HipPrims& S2 = *c->gprims[1];
      Mat PA{sa, 256, (long)an}, PC{sc, 1536, (long)cn}, PD{sc + cn * 32, 1536, (long)cn};
      for (int r = 0; r < 3; ++r) {
        if (side == 1) S2.gemm(32, 1536, 1536, 256, -1.0, PA, PA, 1.0, PC, GEMM_TB | GEMM_TRI_OUT, 0.0);
        else S2.copy(32, 1536, PC, PD);
      }
    }
    c->prims->leaf_stamps = dst;
    c->prims->potrf_leaf(nb, A, inv, (long)n * n, 0, dinfo);
    c->prims->leaf_stamps = nullptr;
    if (side) BAYENS_HIP(hipStreamSynchronize(c->gstreams[1]));
    BAYENS_HIP(hipEventRecord(c->t0, c->stream));
    for (int r = 0; r < reps; ++r) { c->prims->copy(nb, n, S, A); c->prims->potrf_leaf(nb, A, inv, (long)n * n, 0, dinfo); }
    BAYENS_HIP(hipEventRecord(c->t1, c->stream));
    BAYENS_HIP(hipEventSynchronize(c->t1));
    float f = 0;
    BAYENS_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
    if (us_per_launch) *us_per_launch = f * 1e3 / reps;
    if (stamps16) BAYENS_HIP(hipMemcpy(stamps16, dst, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return 0;
  });
}

namespace {
typedef double mfma_v4d __attribute__((ext_vector_type(4)));
// register-only MFMA loop: 16 independent accumulators per wave, no memory traffic - what the matrix pipes sustain
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(int iters, double* sink) {
  mfma_v4d acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = mfma_v4d{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;     // keeps the loop alive
}
}  // namespace

namespace {
// the same stream with the GEMM K loop's operand traffic: per 16 MFMAs, 4 + 4 operands come from LDS (no barriers,
// no global loads) - separates "feeding the pipes from LDS" from everything else in the K loop
// mode bits: 1 = one workgroup barrier per 64-MFMA step, 2 = eight ds_write_b64 per step (the next tile's staging),
// 4 = eight 16-byte global loads per step whose data the writes wait for, 8 = those loads are issued a whole step
// ahead (second register set), 16 = the stores sit in the middle of the step's MFMA stream instead of its end
__global__ __launch_bounds__(256, 2) void mfma_lds_kernel(int iters, double* sink, int mode, const double* src) {
  __shared__ double T[4][16][130];
  for (int e = threadIdx.x; e < 4 * 16 * 130; e += 256) (&T[0][0][0])[e] = 1.0 + e * 1e-9;
  __syncthreads();
  mfma_v4d acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = mfma_v4d{0.0, 0.0, 0.0, 0.0};
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  typedef double v2 __attribute__((ext_vector_type(2)));
  v2 g[8], h[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) { g[q] = v2{1.0, 1.0}; h[q] = v2{1.0, 1.0}; }
  const v2* gp = reinterpret_cast<const v2*>(src) + (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    const int buf = (it & 1) * 2;
    if (mode & 8) {      // what was requested a step ago becomes the data to store; request the next
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = h[q];
#pragma unroll
      for (int q = 0; q < 8; ++q) h[q] = gp[(size_t)q * 256 + ((it & 63) << 16)];
    } else if (mode & 4) {
#pragma unroll
      for (int q = 0; q < 8; ++q) g[q] = gp[(size_t)q * 256 + ((it & 63) << 16)];
    }
    auto stores = [&]() {
      const int kp = threadIdx.x & 7, r = threadIdx.x >> 3;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        T[(buf ^ 2)][2 * kp][r + 32 * q] = g[q].x;
        T[(buf ^ 2)][2 * kp + 1][r + 32 * q] = g[q].y;
        T[(buf ^ 2) + 1][2 * kp][r + 32 * q] = g[4 + q].x;
        T[(buf ^ 2) + 1][2 * kp + 1][r + 32 * q] = g[4 + q].y;
      }
    };
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk == 2 && (mode & 16) && (mode & 2)) stores();
      double a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = T[buf][kk * 4 + lk][wm * 64 + i * 16 + lr];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = T[buf + 1][kk * 4 + lk][wn * 64 + j * 16 + lr];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    if ((mode & 2) && !(mode & 16)) stores();
    if (mode & 1) __syncthreads();
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) sink[0] = s;
}
}  // namespace

int bayens_bench_mfma_lds(bayens_ctx* c, int mode, double* tflops) {
  return guarded(c, [&]() {
    if (!tflops) return fail(c, BAYENS_E_ARG, "bench_mfma_lds: bad argument");
    Stage st(c, BAYENS_DEVICE);
    double* src = st.scratch<double>((size_t)64 * 65536 * 2 + (size_t)512 * 256 * 8 * 2);
    BAYENS_HIP(hipMemsetAsync(src, 0, ((size_t)64 * 65536 * 2 + (size_t)512 * 256 * 8 * 2) * sizeof(double), c->stream));
    hipDeviceProp_t p;
    BAYENS_HIP(hipGetDeviceProperties(&p, c->device));
    const int iters = 5000;
    const int blocks = p.multiProcessorCount * 2;
    double* sink = c->small_scratch(1);
    hipLaunchKernelGGL(mfma_lds_kernel, dim3(blocks), dim3(256), 0, c->stream, 50, sink, mode, src);
    BAYENS_HIP(hipEventRecord(c->t0, c->stream));
    hipLaunchKernelGGL(mfma_lds_kernel, dim3(blocks), dim3(256), 0, c->stream, iters, sink, mode, src);
    BAYENS_HIP(hipEventRecord(c->t1, c->stream));
    BAYENS_HIP(hipEventSynchronize(c->t1));
    float ms = 0;
    BAYENS_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *tflops = (double)blocks * 4 * iters * 64 * (2.0 * 16 * 16 * 4) / (ms * 1e-3) / 1e12;
    return 0;
  });
}

int bayens_bench_mfma_peak(bayens_ctx* c, int waves_per_simd, double* tflops) {
  return guarded(c, [&]() {
    if (waves_per_simd < 1 || waves_per_simd > 2 || !tflops) return fail(c, BAYENS_E_ARG, "bench_mfma_peak: bad argument");
    hipDeviceProp_t p;
    BAYENS_HIP(hipGetDeviceProperties(&p, c->device));
    const int iters = 20000;
    const int blocks = p.multiProcessorCount * waves_per_simd;          // 4 waves per block = one per SIMD
    double* sink = c->small_scratch(1);
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->stream, 100, sink);   // warm-up
    BAYENS_HIP(hipEventRecord(c->t0, c->stream));
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, c->stream, iters, sink);
    BAYENS_HIP(hipEventRecord(c->t1, c->stream));
    BAYENS_HIP(hipEventSynchronize(c->t1));
    float ms = 0;
    BAYENS_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    const double flop = (double)blocks * 4 * iters * 16 * (2.0 * 16 * 16 * 4);
    *tflops = flop / (ms * 1e-3) / 1e12;
    return 0;
  });
}

int bayens_bench_potrf(bayens_ctx* c, int nb, int n, int reps, double* us_per_factor) {
  return guarded(c, [&]() {
    if (n % LEAF || n <= 0 || reps <= 0) return fail(c, BAYENS_E_ARG, "bench_potrf: n must be a positive multiple of 64");
    Stage st(c, BAYENS_DEVICE);
    const size_t an = (size_t)nb * n * n;
    double* dA = st.scratch<double>(an);
    double* dS = st.scratch<double>(an);
    double* inv = st.scratch<double>((size_t)nb * n * LEAF);
    int* dinfo = st.scratch<int>(nb);
    BAYENS_HIP(hipMemsetAsync(dinfo, 0, nb * sizeof(int), c->stream));
    c->prims->fill_random(dA, an, 11);
    Mat A{dA, n, (long)n * n}, S{dS, n, (long)n * n};
    c->prims->gemm(nb, n, n, n, 1.0, A, A, 0.0, S, GEMM_TB, (double)n);   // S = A A^T + n I
    auto timed = [&](bool factor) {
      BAYENS_HIP(hipEventRecord(c->t0, c->stream));
      for (int r = 0; r < reps; ++r) {
        c->prims->tri_copy(nb, n, S, A);
        if (factor) potrf(*c->prims, nb, n, A, inv, (long)n * LEAF, dinfo);
      }
      BAYENS_HIP(hipEventRecord(c->t1, c->stream));
      BAYENS_HIP(hipEventSynchronize(c->t1));
      float f = 0;
      BAYENS_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
      return (double)f * 1e3 / reps;
    };
    timed(true);   // warm-up
    const double with = timed(true), without = timed(false);
    if (us_per_factor) *us_per_factor = with - without;
    return 0;
  });
}

int bayens_bench_trsm(bayens_ctx* c, int nb, int m, int n, int mode, int rhs_lower, int reps, double* us_per_solve) {
  return guarded(c, [&]() {
    if (n % LEAF || m % LEAF || n <= 0 || m <= 0 || reps <= 0 || (rhs_lower && (mode != 1 || m != n)))
      return fail(c, BAYENS_E_ARG, "bench_trsm: m, n positive multiples of 64; rhs_lower needs mode 1 and m == n");
    Stage st(c, BAYENS_DEVICE);
    const size_t an = (size_t)nb * n * n, bn = (size_t)nb * m * n;
    double* dA = st.scratch<double>(an);
    double* dS = st.scratch<double>(an);
    double* dB = st.scratch<double>(bn);
    double* inv = st.scratch<double>((size_t)nb * n * LEAF);
    int* dinfo = st.scratch<int>(nb);
    BAYENS_HIP(hipMemsetAsync(dinfo, 0, nb * sizeof(int), c->stream));
    c->prims->fill_random(dA, an, 11);
    Mat A{dA, n, (long)n * n}, S{dS, n, (long)n * n}, Bm{dB, n, (long)m * n};
    c->prims->gemm(nb, n, n, n, 1.0 / n, A, A, 0.0, S, GEMM_TB, 1.0);   // S = A A^T / n + I
    potrf(*c->prims, nb, n, S, inv, (long)n * LEAF, dinfo);
    if (rhs_lower) c->prims->identity(nb, n, Bm); else c->prims->fill_random(dB, bn, 12);
    auto timed = [&]() {
      BAYENS_HIP(hipEventRecord(c->t0, c->stream));
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) trsm_rlt(*c->prims, nb, m, n, S, inv, (long)n * LEAF, Bm);
        else trsm_rln(*c->prims, nb, m, n, S, inv, (long)n * LEAF, Bm, rhs_lower != 0);
      }
      BAYENS_HIP(hipEventRecord(c->t1, c->stream));
      BAYENS_HIP(hipEventSynchronize(c->t1));
      float f = 0;
      BAYENS_HIP(hipEventElapsedTime(&f, c->t0, c->t1));
      return (double)f * 1e3 / reps;
    };
    timed();   // warm-up
    if (us_per_solve) *us_per_solve = timed();
    return 0;
  });
}

int bayens_test_potrf(bayens_ctx* c, int nb, int n, double* A, int* info) {
  return guarded(c, [&]() {
    if (n % LEAF) return fail(c, BAYENS_E_ARG, "test_potrf: n must be a multiple of 64");
    Stage st(c, BAYENS_HOST);
    const size_t an = (size_t)nb * n * n;
    double* dA = st.scratch<double>(an);
    double* dT = st.scratch<double>(an);
    double* inv = st.scratch<double>((size_t)nb * n * LEAF);
    int* dinfo = st.scratch<int>(nb);
    BAYENS_HIP(hipMemcpyAsync(dA, A, an * sizeof(double), hipMemcpyHostToDevice, c->stream));
    BAYENS_HIP(hipMemsetAsync(dinfo, 0, nb * sizeof(int), c->stream));
    Mat Am{dA, n, (long)n * n}, Tm{dT, n, (long)n * n};
    c->prims->tri_copy(nb, n, Am, Tm);
    potrf(*c->prims, nb, n, Tm, inv, (long)n * LEAF, dinfo);
    BAYENS_HIP(hipMemcpyAsync(A, dT, an * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    BAYENS_HIP(hipMemcpyAsync(info, dinfo, nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    return 0;
  });
}

int bayens_test_trsm(bayens_ctx* c, int nb, int m, int n, const double* L, double* Bm, int mode) {
  return guarded(c, [&]() {
    if ((n % LEAF) || (m % LEAF)) return fail(c, BAYENS_E_ARG, "test_trsm: m, n must be multiples of 64");
    // The solves need the inverted diagonal blocks: refactor L L^T (cheap at test sizes) to obtain them.
    Stage st(c, BAYENS_HOST);
    const size_t ln = (size_t)nb * n * n, bn = (size_t)nb * m * n;
    const double* dL = st.in(L, ln);
    double* dA = st.scratch<double>(ln);
    double* dB = st.scratch<double>(bn);
    double* inv = st.scratch<double>((size_t)nb * n * LEAF);
    int* dinfo = st.scratch<int>(nb);
    BAYENS_HIP(hipMemsetAsync(dinfo, 0, nb * sizeof(int), c->stream));
    BAYENS_HIP(hipMemcpyAsync(dB, Bm, bn * sizeof(double), hipMemcpyHostToDevice, c->stream));
    Mat Lm{const_cast<double*>(dL), n, (long)n * n}, Am{dA, n, (long)n * n}, Bv{dB, n, (long)m * n};
    c->prims->gemm(nb, n, n, n, 1.0, Lm, Lm, 0.0, Am, GEMM_TB | GEMM_TRI_OUT, 0.0);   // A = L L^T (lower blocks)
    c->prims->tri_copy(nb, n, Am, Am);
    potrf(*c->prims, nb, n, Am, inv, (long)n * LEAF, dinfo);
    if (mode == 0) trsm_rlt(*c->prims, nb, m, n, Am, inv, (long)n * LEAF, Bv);
    else trsm_rln(*c->prims, nb, m, n, Am, inv, (long)n * LEAF, Bv, mode == 2);
    BAYENS_HIP(hipMemcpyAsync(Bm, dB, bn * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    st.finish();
    return 0;
  });
}

}  // extern "C"
