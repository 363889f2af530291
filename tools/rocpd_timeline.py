"""Timeline (start, end, duration, queue) of the second half of the kernel dispatches in a rocprofv3 rocpd SQLite
result - for runs that repeat one operation twice.  Usage: rocpd_timeline.py results.db"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end, d.queue_id from rocpd_kernel_dispatch d join "
                  "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
rows = rows[len(rows) // 2:]
t0 = rows[0][1]
for name, st, en, q in rows:
    short = re.sub(r"_ZN6bayens12_GLOBAL__N_1\d+", "", name)[:28]
    print(f"{short:28s} q{q} start {(st - t0) / 1e3:9.1f} end {(en - t0) / 1e3:9.1f} dur {(en - st) / 1e3:8.1f}")
