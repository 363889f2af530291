"""Per-kernel totals of one namespace's kernels in a rocprofv3 rocpd SQLite result.
Usage: rocpd_ns_stats.py results.db bayens_f32 [repeats]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ns = sys.argv[2]
reps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = db.execute("select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d join "
                  "rocpd_info_kernel_symbol s on d.kernel_id = s.id where s.kernel_name like ? group by s.kernel_name "
                  "order by 3 desc", ("%" + ns + "%",)).fetchall()
tot = sum(r[2] for r in rows)
for n, c, t in rows[:24]:
    short = re.sub(r"_ZN\d+" + ns + r"\d+_GLOBAL__N_1\d+", "", n)[:60]
    print(f"{short:60s} {c:5d} {t / reps / 1e6:9.3f} ms {100 * t / tot:5.1f}%")
print("total", tot / reps / 1e6, "ms")
