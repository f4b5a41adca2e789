"""Overlap between the streams of the pipelined duplex step in a rocprofv3 (rocpd sqlite) kernel trace.

    python scripts/rocpd_overlap.py <results.db> [window_ms] > profiles/<name>_overlap.csv

Takes the last `window_ms` (default 40) of the trace before the final 10 ms, lists every kernel with start / end relative to
the window, its queue and stream ids, and summarises per queue: busy time, and how much of it ran while ANOTHER queue was busy.
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    win = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    print("# columns of `kernels`: " + " ".join(cols))
    qcol = "queue_id" if "queue_id" in cols else None
    scol = "stream_id" if "stream_id" in cols else None
    sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0")
    rows = c.execute(f"select {sel} from kernels order by start").fetchall()
    t_end = rows[-1][2] - 10e6
    t0 = t_end - win * 1e6
    rows = [r for r in rows if r[1] >= t0 and r[2] <= t_end]
    print("kernel,start_us,end_us,queue,stream")
    for name, st, en, q, s in rows:
        name = name.split("(")[0].replace("void ", "")[:48]
        print(f'"{name}",{(st - t0) / 1e3:.2f},{(en - t0) / 1e3:.2f},{q},{s}')
    # per-queue busy time and overlapped time (sweep)
    queues = sorted({(r[3], r[4]) for r in rows})
    ev = []
    for name, st, en, q, s in rows:
        ev.append((st, 1, (q, s)))
        ev.append((en, -1, (q, s)))
    ev.sort()
    active = {k: 0 for k in queues}
    busy = {k: 0.0 for k in queues}
    shared = {k: 0.0 for k in queues}
    any_busy = 0.0
    last = ev[0][0]
    for t, d, k in ev:
        dt = (t - last) / 1e3
        live = [q for q in queues if active[q] > 0]
        if live:
            any_busy += dt
        for q in live:
            busy[q] += dt
            if len(live) > 1:
                shared[q] += dt
        active[k] += d
        last = t
    span = (rows[-1][2] - rows[0][1]) / 1e3
    print(f"# window {span:.0f} us, some kernel running {any_busy:.0f} us")
    for q in queues:
        n = sum(1 for r in rows if (r[3], r[4]) == q)
        print(f"# queue {q[0]} stream {q[1]}: {n} kernels, busy {busy[q]:.0f} us, of which {shared[q]:.0f} us while another queue was busy too")


if __name__ == "__main__":
    main()
