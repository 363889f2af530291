"""Build-time check of the generated gfx950 code for ONE code-generation defect of the ROCm 7.2 compiler:

a vector instruction placed at the head of a block that control reaches with EXEC = 0 - ahead of the `s_or_b64 exec, exec, s[..]` that
re-enables the lanes.  Such an instruction writes no lane.  The case that was found (medium.hip built with the computed `Lane`
accessors): the register allocator parks a vector register that must survive a call (`v_mov_b32 v190, v209`, the magic reciprocal of
N) at the top of the join block of a `for (t = tid; t < NP; t += NTH)` loop, behind a rematerialised `s_mov_b32` and in front of the
`s_or_b64 exec`; the copy does nothing, the value restored after the call is whatever the spare register held, `e / N` in the export
of the log-likelihood basis comes out as 8..34 instead of 0/1, and the kernel reads up to 140 tiles below its scratch - a wrong
`basis` for every batch, a memory fault when the scratch is a mapping of its own (20 members).

The check compiles every translation unit of the product to assembly (both precisions) and fails if, in any block entered by
`s_cbranch_execz` or falling out of an `s_cbranch_execnz` loop, a lane-dependent vector instruction precedes the block's
`s_or_b64 exec, exec, ...`.  `__graft_entry__.build()` and tests/test_abi.py run it.

    python tools/check_exec_prologue.py            # the product sources
    python tools/check_exec_prologue.py file.s ... # listings that already exist
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VEC = ("v_", "ds_", "global_", "scratch_", "buffer_", "flat_")
LANEFREE = ("v_readlane", "v_writelane", "v_readfirstlane")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def scan(text):
    """-> [(function, where, line number, instruction)]"""
    lines = text.split("\n")
    n = len(lines)
    execz_targets = set()
    for l in lines:
        m = re.match(r"\s+s_cbranch_execz\s+(\.LBB\d+_\d+)", l)
        if m:
            execz_targets.add(m.group(1))
    out, func = [], None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func = m.group(1)
        m = LABEL.match(l)
        where = None
        if m and m.group(1) in execz_targets:
            where = m.group(1)
        elif re.match(r"\s+s_cbranch_execnz\s", l):
            where = "behind the loop exit at line %d" % (i + 1)
        if where is None:
            continue
        pend = []
        j = i + 1
        while j < n:
            t = lines[j].split(";")[0].strip()
            if not t:
                j += 1
                continue
            if lines[j].startswith(".Lfunc_end"):
                break
            if LABEL.match(lines[j]):          # (a label inside the straight-line code behind a loop exit: keep going only for labels)
                if where.startswith(".LBB"):
                    break
                j += 1
                continue
            op = t.split()[0]
            if re.match(r"s_or_b64\s+exec,\s*exec,", t):
                out += [(func, where, k + 1, p) for k, p in pend]
                break
            if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
                break
            if re.search(r"\bexec\b", t) or "saveexec" in op:
                break
            if op.startswith(VEC) and not op.startswith(LANEFREE):
                pend.append((j, t))
            j += 1
    return out


def product_units():
    sys.path.insert(0, ROOT)
    from bayesian_ensembling_amd import build as b
    return [(f, False) for f in b.SOURCES if f.endswith(".hip")] + [(f, True) for f in b.SOURCES_F32 if f.endswith(".hip")], b


def compile_unit(args):
    src, f32, csrc, arch, tmp, defs = args
    out = os.path.join(tmp, src + (".f32.s" if f32 else ".s"))
    cmd = ["hipcc", f"--offload-arch={arch}", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only"] + \
          (["-DBAYENS_F32"] if f32 else []) + defs + [os.path.join(csrc, src), "-o", out]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def main(argv):
    listings = [a for a in argv if a.endswith(".s")]
    tmp = None
    if not listings:
        units, b = product_units()
        tmp = tempfile.mkdtemp(prefix="bayens_isa_")
        defs = os.environ.get("BAYENS_EXTRA_DEFS", "").split()
        with ThreadPoolExecutor(max_workers=max(1, min(6, os.cpu_count() or 1))) as pool:
            listings = list(pool.map(compile_unit, [(s, f, b.CSRC, b.ARCH, tmp, defs) for s, f in units]))
    bad = 0
    for p in listings:
        seen, r = set(), []
        for hit in scan(open(p).read()):
            if hit[2] not in seen:
                seen.add(hit[2])
                r.append(hit)
        bad += len(r)
        for func, where, line, ins in r:
            print(f"{os.path.basename(p)}:{line}: `{ins}` runs with EXEC = 0 ({where}, {func})")
    print(f"check_exec_prologue: {len(listings)} listing(s), {bad} vector instruction(s) ahead of their block's exec restore")
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
