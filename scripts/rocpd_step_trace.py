"""Launch-by-launch listing of the last complete frame step in a rocprofv3 (rocpd sqlite) kernel trace: kernel, grid,
duration and the idle gap in front of it.  The step is delimited by two consecutive launches of `--mark` (default k_lm_commit).

    python scripts/rocpd_step_trace.py <results.db> [--mark k_lm_commit] > profiles/<name>_step_trace.csv
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    mark = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--mark" else "k_lm_commit"
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if r[0].startswith(mark)]
    if len(idx) < 2:
        sys.exit(f"fewer than two launches of {mark} in {db}")
    a, b = idx[-2], idx[-1]
    print("pos,kernel,workgroups,threads,duration_us,gap_before_us")
    prev_end = rows[a][2]
    busy = 0.0
    for pos, (name, st, en, gx, gy, wx) in enumerate(rows[a + 1:b + 1]):
        name = name.split("(")[0].replace("void ", "")
        wgs = (gx // max(wx, 1)) * max(gy, 1)
        print(f'{pos},"{name}",{wgs},{wx},{(en - st) / 1e3:.2f},{(st - prev_end) / 1e3:.2f}')
        busy += (en - st) / 1e3
        prev_end = en
    span = (rows[b][2] - rows[a][2]) / 1e3
    print(f"# step span {span:.1f} us, kernels busy {busy:.1f} us, idle {span - busy:.1f} us, launches {b - a}")


if __name__ == "__main__":
    main()
