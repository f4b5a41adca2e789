"""Summarise a rocprofv3 (rocpd sqlite) kernel trace as a per-kernel stats CSV (what `--stats` prints).

    python scripts/rocpd_stats.py <results.db> [--header "comment"] > profiles/<name>.csv
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    header = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--header" else None
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    if header:
        print("# " + header)
    print("kernel,calls,total_us,avg_us,min_us,max_us,pct")
    for n, cnt, s, a, mn, mx in rows:
        n = n if len(n) < 200 else n[:197] + "..."
        print(f'"{n}",{cnt},{s/1e3:.1f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*s/tot:.2f}')


if __name__ == "__main__":
    main()
