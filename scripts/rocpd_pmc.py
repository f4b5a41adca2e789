"""Per-kernel average of a rocprofv3 --pmc counter from the rocpd sqlite output.

    python scripts/rocpd_pmc.py <results.db> [--header "comment"]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.  On gfx950 FETCH_SIZE counts a wide coalesced streaming read at
exactly half its bytes (128-byte requests tallied at 64 B; /opt/skills/guides/MI355X_MICROARCH.md, section HBM), so the
`bytes_corrected` column doubles it; WRITE_SIZE is reported as is (uncalibrated, see the same section).
"""
import os
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
    for name, ctr, n, avg, mn, mx, dur in rows[:int(os.environ.get("PMC_ROWS", "12"))]:
        name = name if len(name) < 160 else name[:157] + "..."
        corr = avg * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
        print(f'"{name}",{ctr},{n},{avg:.1f},{mn:.1f},{mx:.1f},{corr:.0f},{dur/1e3:.2f}')
    # One kernel name serves several GEMM shapes (k_gemm_xlds: in_proj, linear_in, the text head): `--clusters <substring>` lists
    # that kernel's dispatches grouped by counter value (bins of 4 MiB), i.e. one line per shape it was launched on.
    if "--clusters" in sys.argv:
        sub = sys.argv[sys.argv.index("--clusters") + 1]
        vals = c.execute("select kernel_name, counter_name, value, duration from counters_collection where kernel_name like ?",
                         (f"%{sub}%",)).fetchall()
        bins = {}
        by_dur = "--by-duration" in sys.argv       # cycle / request counters: group by the dispatch's duration (bins of 4 us) instead
        for name, ctr, v, dur in vals:
            key = (name.split("(")[0].replace("void ", ""), ctr, int(dur // 4000) if by_dur else int(v // 4096))
            b = bins.setdefault(key, [0, 0.0, 0.0])
            b[0] += 1; b[1] += v; b[2] += dur
        print("# clusters: kernel,counter,dispatches,avg_KiB,bytes_corrected,avg_duration_us_under_pmc")
        for (name, ctr, _), (n, sv, sd) in sorted(bins.items(), key=lambda kv: -kv[1][0]):
            if n < 4:
                continue
            corr = sv / n * 1024 * (2 if ctr == "FETCH_SIZE" else 1)
            print(f'"{name}",{ctr},{n},{sv / n:.1f},{corr:.0f},{sd / n / 1e3:.2f}')


if __name__ == "__main__":
    main()
