# (needs scripts/experiments/mimi_tr_grouped.patch applied: MMI_MIMI_TR is read by the patched engine only)
# per-phase kernel time of the grouped Mimi transformer in its one-launch-per-phase form (MMI_MIMI_TR=phases): the kernel trace's
# k_mimi_tr dispatches, averaged by position inside a transformer (41 phases: gather, then 8 x [in_proj, attention, out_proj, linear1, linear2])
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/trprof; mkdir -p $O/trprof
MMI_MIMI_TR=phases timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trprof -- python bench.py --workload mimi --batch ${1:-8} --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $O/trprof/bench.log 2>&1
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/trprof"
f = glob.glob(O + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tr = [r for r in rows if r["Kernel_Name"].startswith("void k_mimi_tr") or "k_mimi_tr" in r["Kernel_Name"]]
print("k_mimi_tr dispatches:", len(tr))
# group consecutive runs of k_mimi_tr
runs, cur, prev_idx = [], [], None
idx = {id(r): i for i, r in enumerate(rows)}
for r in tr:
    i = idx[id(r)]
    if prev_idx is not None and i != prev_idx + 1:
        runs.append(cur); cur = []
    cur.append(r); prev_idx = i
runs.append(cur)
runs = [x for x in runs if len(x) == 41]
print("complete transformers:", len(runs))
acc = collections.defaultdict(list); gap = collections.defaultdict(list)
for run in runs[len(runs)//3:]:
    for p, r in enumerate(run):
        acc[p].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if p: gap[p].append((int(r["Start_Timestamp"]) - int(run[p-1]["End_Timestamp"])) / 1e3)
names = ["gather"] + ["L%d.%s" % (l, s) for l in range(8) for s in ("in_proj", "attn", "out_proj", "linear1", "linear2")]
tot = 0
with open(O + "/../tr_phases_profile.txt", "w") as out:
    for p in range(41):
        d = sum(acc[p]) / len(acc[p]); g = sum(gap[p]) / len(gap[p]) if p else 0.0
        tot += d + g
        line = "%-14s kernel %6.2f us   gap before %5.2f us" % (names[p], d, g)
        print(line); out.write(line + "\n")
    line = "one transformer: %.1f us (kernels + gaps)" % tot
    print(line); out.write(line + "\n")
PY
