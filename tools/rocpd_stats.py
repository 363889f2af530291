"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into the per-kernel table rocprofv3 --stats prints:
calls, total / average / min / max duration, share of GPU time.  Usage: rocpd_stats.py results.db [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)[:80]


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "min(d.start), max(d.end) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
        "on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = max(r[6] for r in rows) - min(r[5] for r in rows)
    lines = [f"# rocprofv3 --kernel-trace summary of {sys.argv[1]}",
             f"# total kernel time {total / 1e6:.3f} ms over a {span / 1e6:.3f} ms span ({100.0 * total / span:.1f}% busy)",
             f"{'kernel':80s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}"]
    for name, calls, tot, mn, mx, _, _ in rows:
        lines.append(f"{short(name):80s} {calls:7d} {tot / 1e6:10.3f} {tot / calls / 1e3:10.2f} {mn / 1e3:9.2f} "
                     f"{mx / 1e3:10.2f} {100.0 * tot / total:6.2f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
