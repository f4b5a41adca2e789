"""Per-SITE kernel time of the frame step from a rocprofv3 (rocpd sqlite) kernel trace.

Several GEMM shapes share one kernel name (k_gemm_xp serves in_proj, out_proj, linear_out, the text head and the depth
transformer's linears), so a per-kernel summary cannot give per-GEMM roofline fractions.  The engine records which site of the
step issued which kernel (`mmi_lm_launch_list` / `mmi_mimi_launch_list`, one "site<TAB>kernel" line per launch, dumped by
`bench.py --launch-lists DIR`); a step's launches appear in the trace in exactly that order, so the two are joined by position.

    python scripts/rocpd_sites.py <results.db> <launch_list_dir> [--header "comment"] > profiles/<name>_sites.csv

Columns: program, site, launches per step, mean us per step, mean us per launch, algorithmic MB per launch (weight-streaming
sites of the 7B bf16 model; blank otherwise), GB/s, fraction of the 8 TB/s HBM peak.
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
    out = []
    for line in Path(path).read_text().splitlines():
        if line.strip():
            site, _, kern = line.partition("\t")
            out.append((site, kern))
    return out


def main():
    db, ldir = sys.argv[1], Path(sys.argv[2])
    header = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--header" else None
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
    names = [short(r[0]) for r in rows]
    dur = [(r[2] - r[1]) / 1e3 for r in rows]
    if header:
        print("# " + header)
    print("program,site,launches_per_step,us_per_step,us_per_launch,algorithmic_MB_per_launch,GBps,frac_of_8TBps")
    nbytes = site_bytes_7b()
    for prog in ("lm", "mimi_encode", "mimi_decode"):
        f = ldir / f"launch_list_{prog}.tsv"
        if not f.exists():
            continue
        ll = load_list(f)
        want = [k for _, k in ll]
        n = len(want)
        starts = [i for i in range(len(names) - n + 1) if names[i] == want[0] and names[i:i + n] == want]
        # keep the graph-replayed steps of the timed region: the last half of the matches
        starts = starts[len(starts) // 2:]
        if not starts:
            print(f"# {prog}: no step of {n} launches found in the trace", file=sys.stderr)
            continue
        per_site, cnt, order = {}, {}, []
        for st in starts:
            for j, (site, _) in enumerate(ll):
                if site not in per_site:
                    per_site[site] = 0.0
                    cnt[site] = 0
                    order.append(site)
                per_site[site] += dur[st + j]
        for site, _ in ll:
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
