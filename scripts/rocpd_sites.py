"""Per-SITE kernel time of the frame step from a rocprofv3 (rocpd sqlite) kernel trace.

Several GEMM shapes share one kernel name (k_gemm_xp serves in_proj, out_proj, linear_out, the text head and the depth
transformer's linears), so a per-kernel summary cannot give per-GEMM roofline fractions.  The engine records which site of the
step issued which kernel (`mmi_lm_launch_list` / `mmi_mimi_launch_list`, one "site<TAB>kernel" line per launch, dumped by
`bench.py --launch-lists DIR`); a step's launches appear in the trace in exactly that order, so the two are joined by position.

    python scripts/rocpd_sites.py <results.db> <launch_list_dir> [--header "comment"] [--last N] > profiles/<name>_sites.csv

--last N: average the LAST N steps of the trace instead of its last half.  bench.py moves the sessions to their mid-run ring depth
only after the staggered start (248 steps at 32 sessions), so "the last half" of a default run still holds shallow, partly masked
steps; the steps after the seek are warm-up + timed + latency + profile passes (>= 120 with the default flags).

Columns: program, site, launches per step, mean us per step, mean us per launch, algorithmic MB per launch (the packed weight
bytes the engine recorded for the launch - third column of the launch list, right for bf16, int8 and fp8 weights alike; the
7B bf16 table below only serves lists written before the engine recorded them), GB/s, fraction of the 8 TB/s HBM peak.
The pipelined step runs its three programs on three streams: each program is matched on the stream that replays it.
"""
import sqlite3
import sys
from pathlib import Path

HBM_PEAK_GBS = 8000.0


def site_bytes_7b():
    """Packed weight bytes one launch of each weight-streaming site reads (Moshi-7B, bf16): SURVEY.md 8(a) shapes."""
    d, h, dd, dh, V, card, q = 4096, 11264, 1024, 2816, 32000, 2048, 8
    return {
        "L.in_proj": 2 * 3 * d * d, "L.out_proj": 2 * d * d, "L.ffn_in": 2 * 2 * h * d, "L.ffn_out": 2 * d * h,
        "text_linear": 2 * V * d, "dep.in_all": 2 * q * dd * d, "dep.in_proj": 2 * 3 * dd * dd, "dep.out_proj": 2 * dd * dd,
        "dep.attn_out_proj": 2 * dd * dd,
        "dep.ffn_in": 2 * 2 * dh * dd, "dep.ffn_out": 2 * dd * dh, "dep.lin": 2 * card * dd,
    }


def short(name):
    name = name.replace("void ", "").strip()
    for cut in "<(":
        i = name.find(cut)
        if i >= 0:
            name = name[:i]
    return name.strip()


def load_list(path):
    """[(site, kernel, weight bytes or 0)]: `bench.py --launch-lists` writes the engine's launch list, whose GEMM lines carry the
    packed weight bytes of the launch (whatever the weight format: bf16, int8, fp8)."""
    out = []
    for line in Path(path).read_text().splitlines():
        if line.strip():
            f = line.split("\t")
            out.append((f[0], f[1] if len(f) > 1 else "", int(f[2]) if len(f) > 2 else 0))
    return out


def main():
    db, ldir = sys.argv[1], Path(sys.argv[2])
    header = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--header" else None
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 0
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    # the pipelined step runs the encoder, the LM and the decoder on three streams: a program's launches are consecutive on ITS
    # stream, not in the global order
    scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    rows = c.execute(f"select {name_col}, start, end, {scol} from kernels order by start").fetchall()
    by_stream = {}
    for r in rows:
        by_stream.setdefault(r[3], []).append(r)
    if header:
        print("# " + header)
    print("program,site,launches_per_step,us_per_step,us_per_launch,algorithmic_MB_per_launch,GBps,frac_of_8TBps")
    nbytes = site_bytes_7b()           # fallback for launch lists written before the engine recorded bytes
    seen_bytes = set()
    for prog in ("lm", "mimi_encode", "mimi_decode"):
        f = ldir / f"launch_list_{prog}.tsv"
        if not f.exists():
            continue
        ll = load_list(f)
        want = [k for _, k, _ in ll]
        n = len(want)
        names, dur, starts = [], [], []
        for srows in by_stream.values():           # the stream that replays this program
            nm = [short(r[0]) for r in srows]
            st = [i for i in range(len(nm) - n + 1) if nm[i] == want[0] and nm[i:i + n] == want]
            if len(st) > len(starts):
                names, dur, starts = nm, [(r[2] - r[1]) / 1e3 for r in srows], st
        for site, _, nb in ll:                      # bytes of a site = its heaviest launch (the engine's own figure)
            if nb:
                nbytes[site] = max(nb, nbytes.get(site, 0)) if site in seen_bytes else nb
                seen_bytes.add(site)
        # keep the graph-replayed steps of the timed region: the last half of the matches
        starts = starts[-last:] if last > 0 else starts[len(starts) // 2:]
        if not starts:
            print(f"# {prog}: no step of {n} launches found in the trace", file=sys.stderr)
            continue
        per_site, cnt, order = {}, {}, []
        for st in starts:
            for j, (site, _, _) in enumerate(ll):
                if site not in per_site:
                    per_site[site] = 0.0
                    cnt[site] = 0
                    order.append(site)
                per_site[site] += dur[st + j]
        for site, _, _ in ll:
            cnt[site] += 1
        total = 0.0
        for site in order:
            us_step = per_site[site] / len(starts)
            us_launch = us_step / cnt[site]
            total += us_step
            b = nbytes.get(site) if prog == "lm" else None
            if b:
                gbs = b / (us_launch * 1e-6) / 1e9
                print(f"{prog},{site},{cnt[site]},{us_step:.1f},{us_launch:.2f},{b / 1e6:.1f},{gbs:.0f},{gbs / HBM_PEAK_GBS:.3f}")
            else:
                print(f"{prog},{site},{cnt[site]},{us_step:.1f},{us_launch:.2f},,,")
        print(f"{prog},TOTAL ({len(starts)} steps averaged),{n},{total:.1f},,,,")


if __name__ == "__main__":
    main()
