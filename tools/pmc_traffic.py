"""HBM traffic per launch of the dominant GEMM kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd
SQLite).  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950.
Usage: pmc_traffic.py fetch.db write.db out.json"""
import json
import sqlite3
import sys

KERNEL = "gemm_f64_mfma<false, true>"


def total(dbfile, ctr):
    db = sqlite3.connect(dbfile)
    rows = db.execute("select count(*), sum(value) from counters_collection where counter_name = ? and "
                      "kernel_name like ?", (ctr, "%" + KERNEL + "%")).fetchone()
    return rows[0], rows[1]


nf, fetch_kb = total(sys.argv[1], "FETCH_SIZE")
nw, write_kb = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
fetch_b, write_b = fetch_kb * 1024.0 * 2.0, write_kb * 1024.0
out = {
    "kernel": "bayens::" + KERNEL,
    "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
               "--no-cpu-baseline (separate passes)",
    "launches": nf,
    "fetch_size_kb_raw": fetch_kb,
    "write_size_kb_raw": write_kb,
    "fetch_bytes_corrected_x2": fetch_b,
    "write_bytes": write_b,
    "hbm_bytes_per_launch": (fetch_b + write_b) / nf,
    "calibration": "scale_rows_kernel / kmat_kernel<true> read 1 GiB per call and report 0.50 GiB FETCH_SIZE (the x2 "
                   "correction of MI355X_MICROARCH.md holds for the 8- and 16-byte-per-lane reads used here); "
                   "copy_kernel writes 1 GiB and reports 1048576 KB WRITE_SIZE",
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
