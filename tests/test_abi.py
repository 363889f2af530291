"""The C-ABI library loads on a host without a GPU and exports every symbol include/bayens.h declares;
without a device the product fails loudly (no CPU fallback)."""
import ctypes
import os

import pytest

from bayesian_ensembling_amd import _lib


def _have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="module", autouse=True)
def _built():
    """hipcc cross-compiles gfx950 without a GPU: build the library when the snapshot does not carry it yet."""
    if not os.path.exists(_lib.LIB_PATH):
        from bayesian_ensembling_amd import build as b
        b.build(verbose=False)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libbayens.so is not built (python -m bayesian_ensembling_amd.build)"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.declared_symbols()
    assert len(names) >= 28
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_version_string():
    lib = _lib.load()
    assert b"gfx950" in lib.bayens_version()


@pytest.mark.skipif(_have_gpu(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_fallback():
    with pytest.raises(_lib.BayensError):
        _lib.Context(0)
    import numpy as np
    import bayesian_ensembling_amd as es
    from bayesian_ensembling_amd.labelled import DataArray
    _lib._default_ctx = None
    da = DataArray(np.random.rand(3, 8), ("realisation", "time"),
                   {"realisation": np.arange(3), "time": np.arange(8)})
    mc = es.ModelCollection([es.ProcessModel(da, "a"), es.ProcessModel(da, "b")])
    with pytest.raises(_lib.BayensError):
        mc.fit(es.GPDTW1D(), n_optim_nits=1)


def test_product_never_imports_the_oracle():
    pkg = os.path.dirname(_lib.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "gp_oracle" not in txt and "hostcheck" not in txt.replace("oracle/hostcheck", ""), f


def test_nothing_but_the_gemms_own_dma_setup_touches_m0():
    """tools/check_m0.py: compiles csrc/gemm.hip to gfx950 assembly (no GPU needed) and checks the inline-assembly contract
    of lds_dma16 - the compiler ignores the `m0` clobber, so the absence of compiler-generated M0 traffic is verified."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_m0", os.path.join(root, "tools", "check_m0.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
    bad, _ = mod.check("s_mov_b32 m0, s4\ns_nop 0\nglobal_load_lds_dwordx4 v[0:1], off\nv_readlane_b32 s5, v1, m0\n")
    assert bad and "outside lds_dma16" in bad[0]


def test_no_kernel_of_the_build_carries_a_vector_instruction_ahead_of_its_blocks_exec_restore():
    """tools/check_exec_prologue.py: every translation unit of the product compiled to gfx950 assembly (both precisions, no GPU
    needed) and scanned for the code-generation defect behind round 4's device fault - and the scanner itself
    on the block that faulted, reduced to its skeleton, and on the two legitimate shapes it must not flag."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_exec_prologue", os.path.join(root, "tools", "check_exec_prologue.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main([]) == 0
    faulting = ("_Zkernel:\n\ts_and_saveexec_b64 s[0:1], vcc\n\ts_cbranch_execz .LBB4_39\n.LBB4_38:\n\tds_write_b64 v0, v[200:201]\n"
                "\ts_andn2_b64 exec, exec, s[2:3]\n\ts_cbranch_execnz .LBB4_38\n.LBB4_39:\n\ts_mov_b32 s80, 0xfee00000\n"
                "\tv_mov_b32_e32 v190, v209\n\ts_mov_b32 s81, 0xbfe62e42\n\ts_or_b64 exec, exec, s[0:1]\n\ts_swappc_b64 s[30:31], s[0:1]\n")
    hits = mod.scan(faulting)
    assert hits and all("v190, v209" in h[3] for h in hits)
    clean = faulting.replace("\tv_mov_b32_e32 v190, v209\n\ts_mov_b32 s81, 0xbfe62e42\n\ts_or_b64 exec, exec, s[0:1]\n",
                             "\ts_mov_b32 s81, 0xbfe62e42\n\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v190, v209\n")
    assert mod.scan(clean) == []
    if_else = ("_Zkernel:\n\ts_and_saveexec_b64 s[0:1], vcc\n\ts_xor_b64 s[0:1], exec, s[0:1]\n\ts_cbranch_execz .LBB1_130\n"
               "\tv_lshlrev_b32_e32 v3, 4, v162\n.LBB1_130:\n\ts_andn2_saveexec_b64 s[0:1], s[0:1]\n\tv_xad_u32 v3, v162, v137, v161\n"
               "\ts_or_b64 exec, exec, s[0:1]\n")
    assert mod.scan(if_else) == []
