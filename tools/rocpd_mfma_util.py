"""Duration-weighted MfmaUtil per kernel from a rocprofv3 --pmc MfmaUtil pass (rocpd SQLite).
MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE * number of SIMDs) * 100, rocprofv3's derived metric.
Usage: rocpd_mfma_util.py results.db [out.txt]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, count(*), sum(value * (end - start)) / sum(end - start), sum(end - start) "
                  "from counters_collection where counter_name = 'MfmaUtil' group by kernel_name order by 4 desc").fetchall()
lines = [f"# MfmaUtil (% of cycles the matrix pipes are busy, duration-weighted over the launches) from {sys.argv[1]}",
         f"{'kernel':64s} {'calls':>6s} {'MfmaUtil%':>10s} {'total_ms':>10s}"]
for name, calls, util, dur in rows:
    if util < 0.5:
        continue
    short = re.sub(r"\(.*\)$", "", re.sub(r"^void ", "", re.sub(r"\(anonymous namespace\)::", "", name)))[:64]
    lines.append(f"{short:64s} {calls:6d} {util:10.2f} {dur / 1e6:10.3f}")
out = "\n".join(lines) + "\n"
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out)
print(out)
