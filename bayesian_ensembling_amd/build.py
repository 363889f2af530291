"""Builds libbayens.so (HIP, gfx950) in-tree with hipcc.  `python -m bayesian_ensembling_amd.build`.

Variant builds never replace the product library: with `BAYENS_EXTRA_DEFS` (extra -D flags, e.g. the bounds-checked
mid-size kernel `-DBAYENS_MEDIUM_CHECK`) the output is `libbayens_<tag>.so` (`BAYENS_LIB_TAG`, default "debug"), built in
its own object directory, and the flags it was built with are recorded beside it (`<lib>.defs`) so that a stale variant is
rebuilt when they change.  A process uses a variant only when `BAYENS_LIB` names it (`_lib.LIB_PATH`)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbayens.so")
SOURCES = ["gemm.hip", "kernels.hip", "dba.hip", "small.hip", "medium.hip", "potrf256.hip", "capi.hip", "driver.cpp"]
# the fp32 instance of the fit: the same sources compiled a second time with real = float (namespace bayens_f32)
SOURCES_F32 = ["gemm.hip", "kernels.hip", "small.hip", "driver.cpp", "fit_f32.hip"]
ARCH = "gfx950"


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def variant_path(tag: str) -> str:
    return os.path.join(HERE, f"libbayens_{tag}.so")


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libbayens.so cannot be built on this host")
    defs = os.environ.get("BAYENS_EXTRA_DEFS", "").split()
    tag = os.environ.get("BAYENS_LIB_TAG", "debug") if defs else ""
    lib = variant_path(tag) if defs else LIB
    stamp = lib + ".defs"
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", h) for h in ("bayens.h", "bayens_diag.h")]
    recorded = open(stamp).read().split() if os.path.exists(stamp) else []
    if not force and os.path.exists(lib) and recorded == defs and not any(_newer(d, lib) for d in deps):
        return lib
    bdir = os.path.join(HERE, "build", tag) if defs else os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    jobs = []
    for src, f32 in [(f, False) for f in SOURCES] + [(f, True) for f in SOURCES_F32]:
        obj = os.path.join(bdir, src + (".f32.o" if f32 else ".o"))
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-x", "hip"] + \
              (["-DBAYENS_F32"] if f32 else []) + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
        jobs.append((cmd, obj))

    def compile_one(job):
        if verbose:
            print(" ".join(job[0]), flush=True)
        subprocess.check_call(job[0])
        return job[1]

    # the translation units are independent: compile them side by side (bounded: hipcc peaks at ~2 GB per unit)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(6, os.cpu_count() or 1))) as pool:
        objs = list(pool.map(compile_one, jobs))
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(" ".join(defs))
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
