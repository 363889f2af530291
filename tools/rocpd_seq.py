"""Dispatch sequence of the last `count` kernels in a rocprofv3 rocpd SQLite result: name, duration, gap to the
previous kernel's end.  Usage: rocpd_seq.py results.db [count]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                  "on d.kernel_id = s.id order by d.start").fetchall()
count = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[-count:]
prev = rows[0][1]
tot = {}
for name, st, en in rows:
    short = re.sub(r"\(.*$", "", re.sub(r"^void |bayens::|\(anonymous namespace\)::", "", name))[:48]
    print(f"{short:48s} {(en - st) / 1e3:9.2f} us   gap {(st - prev) / 1e3:7.2f} us")
    t = tot.setdefault(short, [0, 0.0]); t[0] += 1; t[1] += (en - st) / 1e3
    prev = en
print("---- span %.1f us, busy %.1f us" % ((rows[-1][2] - rows[0][1]) / 1e3, sum(v[1] for v in tot.values())))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:48s} x{v[0]:4d} {v[1]:10.1f} us")
