"""Build-time check of the one inline-assembly contract in the product (csrc/gemm.hip: lds_dma16).

The fp64 GEMM loads its operand tiles straight into LDS with `s_mov_b32 m0, <lds offset>; s_nop 0;
global_load_lds_dwordx4`.  clang ignores `m0` in an asm clobber list ("clobber list contains reserved registers"), so
nothing in the language stops the compiler from keeping a value of its own in M0 across the statement.  gfx950 code that
the compiler generates for this translation unit has no reason to touch M0 (LDS instructions stopped needing it with
gfx9) - this script makes that a checked fact instead of an assumption: it compiles gemm.hip to gfx950 assembly and
fails if
  * any instruction mentions m0 other than the asm's own `s_mov_b32 m0, s<N>`,
  * such a move is not followed by `s_nop 0` and `global_load_lds_dwordx4` (i.e. it is not ours),
  * a `global_load_lds_*` appears without the move in front of it, or
  * an instruction that reads M0 implicitly (s_movrel*, v_movrel*, s_sendmsg*, ds_gws_*, v_interp*, buffer_load ... lds,
    s_load/indexed forms are not among them) occurs anywhere in the unit.
Runs on the CPU (hipcc cross-compiles), ~3 s.  Called by __graft_entry__.build() and tests/test_abi.py."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "bayesian_ensembling_amd", "csrc", "gemm.hip")
IMPLICIT_M0 = re.compile(r"^\s*(s_movrel\w*|v_movrel\w*|s_sendmsg\w*|ds_gws_\w+|v_interp\w*|s_ttrace\w*|buffer_load\w*\s.*\blds\b)")


def instructions(asm_text):
    for line in asm_text.splitlines():
        code = line.split(";")[0].rstrip()
        if not code.strip() or code.lstrip().startswith((".", "//")) or code.rstrip().endswith(":"):
            continue
        yield code.strip()


def check(asm_text):
    ins = list(instructions(asm_text))
    problems, n_dma = [], 0
    for i, code in enumerate(ins):
        if IMPLICIT_M0.match(code):
            problems.append(f"instruction with an implicit M0 operand: {code}")
        if re.search(r"\bm0\b", code):
            if not re.fullmatch(r"s_mov_b32 m0, s\d+", code):
                problems.append(f"M0 touched outside lds_dma16: {code}")
            elif not (i + 2 < len(ins) and ins[i + 1] == "s_nop 0" and ins[i + 2].startswith("global_load_lds_dwordx4")):
                problems.append(f"`{code}` is not followed by s_nop 0 + global_load_lds_dwordx4: {ins[i + 1:i + 3]}")
        if code.startswith("global_load_lds"):
            n_dma += 1
            if not (i >= 2 and ins[i - 1] == "s_nop 0" and re.fullmatch(r"s_mov_b32 m0, s\d+", ins[i - 2])):
                problems.append(f"`{code}` without its own M0 set-up in front: {ins[max(0, i - 2):i]}")
    if n_dma == 0:
        problems.append("no global_load_lds instruction found: the check no longer looks at the kernel it was written for")
    return problems, n_dma


def main():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "gemm.s")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S", SRC,
                        "-o", out], check=True, stderr=subprocess.DEVNULL)
        problems, n_dma = check(open(out).read())
    if problems:
        print("M0 check FAILED:\n  " + "\n  ".join(problems[:20]), file=sys.stderr)
        return 1
    print(f"M0 check ok: {n_dma} global_load_lds_dwordx4, each with its own s_mov_b32 m0; nothing else touches M0")
    return 0


if __name__ == "__main__":
    sys.exit(main())
