"""Register / spill / private-segment figures of every kernel of the product, read from the code-object metadata of an
assembly build (no GPU): python tools/kernel_resources.py [out.txt].  One line per kernel; functions that are called, not
launched (`__noinline__`), are listed with the figures of their own body."""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bayesian_ensembling_amd import build as b  # noqa: E402
from tools.check_exec_prologue import compile_unit  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [re.sub(r"\((?!anonymous).*", "", o) for o in out]


def main():
    units = [(f, False) for f in b.SOURCES if f.endswith(".hip")] + [(f, True) for f in b.SOURCES_F32 if f.endswith(".hip")]
    tmp = tempfile.mkdtemp(prefix="bayens_res_")
    with ThreadPoolExecutor(max_workers=6) as pool:
        listings = list(pool.map(compile_unit, [(s, f, b.CSRC, b.ARCH, tmp, []) for s, f in units]))
    rows = []
    for path in listings:
        txt = open(path).read()
        unit = os.path.basename(path)[:-2]
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, flags=re.S):
            blk = m.group(0)
            get = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            rows.append((unit, name, get("vgpr_count"), get("agpr_count"), get("sgpr_count"), get("vgpr_spill_count"), get("sgpr_spill_count"),
                         get("private_segment_fixed_size"), get("group_segment_fixed_size")))
        for m in re.finditer(r"\.set \.L(\S+)\.num_vgpr, (\d+)\n.*?\.set \.L\1\.private_seg_size, (\d+)", txt, flags=re.S):
            if re.search(r"^\s+\.name:\s+" + re.escape(m.group(1)) + r"$", txt, flags=re.M):
                continue
            rows.append((unit, m.group(1) + " [called]", int(m.group(2)), 0, -1, -1, -1, int(m.group(3)), 0))
    names = demangle([r[1].replace(" [called]", "") for r in rows])
    lines = ["%-14s %-78s %5s %5s %5s %7s %7s %8s %8s" % ("unit", "kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "private", "lds-fix")]
    for r, nm in zip(rows, names):
        nm = nm + (" [called]" if r[1].endswith("[called]") else "")
        lines.append("%-14s %-78s %5d %5d %5s %7s %7s %8d %8d" % (r[0], nm[:78], r[2], r[3], r[4] if r[4] >= 0 else "-", r[5] if r[5] >= 0 else "-",
                                                               r[6] if r[6] >= 0 else "-", r[7], r[8]))
    spilling = [l for l in lines[1:] if not re.search(r"\s0\s+0\s+0\s+\d+$", l) and "[called]" not in l]
    lines.append("")
    lines.append("%d kernels, %d with spills or a private segment" % (sum("[called]" not in l for l in lines[1:-1]), len(spilling)))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out + "\n")


if __name__ == "__main__":
    main()
