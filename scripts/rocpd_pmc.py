"""Per-kernel average of a rocprofv3 --pmc counter from the rocpd sqlite output.

    python scripts/rocpd_pmc.py <results.db> [--header "comment"]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read at
exactly half its bytes (128-byte requests tallied at 64 B; /opt/skills/guides/MI355X_MICROARCH.md, section HBM), so the
`bytes_corrected` column doubles it; WRITE_SIZE is reported as is (uncalibrated, see the same section).
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    header = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--header" else None
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                     "from counters_collection group by kernel_name, counter_name order by 4 desc").fetchall()
    if header:
        print("# " + header)
    print("kernel,counter,dispatches,avg_KiB,min_KiB,max_KiB,bytes_corrected,avg_duration_us_under_pmc")
    for name, ctr, n, avg, mn, mx, dur in rows:
        name = name if len(name) < 160 else name[:157] + "..."
        corr = avg * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        print(f'"{name}",{ctr},{n},{avg:.1f},{mn:.1f},{mx:.1f},{corr:.0f},{dur/1e3:.2f}')


if __name__ == "__main__":
    main()
