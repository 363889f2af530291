"""Per-kernel totals of one PMC counter from a rocprofv3 rocpd SQLite result (view counters_collection).
Usage: rocpd_pmc.py results.db COUNTER [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", name))[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    ctr = sys.argv[2]
    rows = db.execute("select kernel_name, count(*), sum(value), sum(end - start) from counters_collection "
                      "where counter_name = ? group by kernel_name order by 3 desc", (ctr,)).fetchall()
    lines = [f"# {ctr} per kernel from {sys.argv[1]} (value unit as rocprofv3 reports it: KB for FETCH_SIZE/WRITE_SIZE)",
             f"{'kernel':70s} {'calls':>6s} {'sum':>16s} {'per_call':>14s} {'total_ms':>10s}"]
    for name, calls, tot, dur in rows:
        lines.append(f"{short(name):70s} {calls:6d} {tot:16.1f} {tot / calls:14.1f} {dur / 1e6:10.3f}")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
