"""Which launch of the C5 step is not reproducible?  (debug tool; MMI_DEBUG_TRACE - see lm_engine.hip)

    MMI_DEBUG_TRACE=/tmp/trace python tests/tools/c5_trace_probe.py [sessions]

Runs the 64-session int8 x int8 case of tests (2 temporal layers at the 7B widths, masks + a partial reset, 2 steps) several times
on ONE handle with the same inputs; the engine checksums every allocation of its streaming state after every op of the launch
list.  Prints, per session, the first (step, op, site, allocation) whose checksum differs from session 0's."""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from moshi_amd.config import LMConfig          # noqa: E402
from moshi_amd.weights import quantize_lm_state_dict, random_lm_state_dict      # noqa: E402
from oracle.lm_oracle import LMOracle           # noqa: E402
from tests import lm_cases                       # noqa: E402

NAMES = ["exec", "offsets", "cache", "user_i32", "tokens", "text_tok", "audio_tok", "out_i32", "x", "xn", "qrot", "att", "hb", "tout",
         "text_logits", "kc", "vc", "opart", "ml", "attn_done", "partial", "rope", "xnq", "attq", "hbq", "toutq", "dxnq", "sx_xn", "sx_tout",
         "sx_dxn", "sx_att", "sx_hb", "dx", "dxn", "dqkv", "datt", "dhb", "dlogits", "dpre", "dkc", "dvc", "noise", "use_noise", "forced",
         "use_forced", "rng"]


def main():
    prefix = os.environ["MMI_DEBUG_TRACE"]
    n_sessions = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    quant = os.environ.get("PROBE_QUANT", "q8")
    B, S, seed = 64, 2, 364
    cfg = LMConfig(num_layers=2, context=64)
    sd = random_lm_state_dict(cfg, seed=seed)
    if quant == "q8":
        sd = quantize_lm_state_dict(sd)
    gen = lm_cases.make_engine(cfg, sd, "cuda", None, B, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg)
    orc.streaming(B)
    rng = np.random.default_rng(seed)
    plan = []
    for s in range(S):
        mask = rng.random(B) > 0.3
        mask[0] = True
        reset = None
        if s == S // 2:
            reset = np.zeros(B, bool); reset[B - 1] = True
            mask[B - 1] = True
        codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
        if reset is not None:
            orc.reset_streaming(reset)
        orc.set_exec_mask(mask)
        oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
        plan.append(dict(mask=mask, reset=reset, codes=codes, forced=np.concatenate([ott[:, None], oat], 1)))
    for _ in range(n_sessions):
        with gen.streaming(B):
            for p in plan:
                if p["reset"] is not None:
                    gen.reset_streaming(torch.from_numpy(p["reset"]).to("cuda"))
                gen.set_exec_mask(torch.from_numpy(p["mask"]).to("cuda"))
                gen.step_with_taps(torch.from_numpy(p["codes"]).to("cuda"), forced_tokens=torch.from_numpy(p["forced"]).to("cuda"))
    dump = os.environ.get("MMI_DEBUG_DUMP")
    if dump:                     # element-level view of the dumped allocations: which (row, column) differ between sessions
        allocs = [int(a) for a in dump.split(":")[2].split(",")]
        for a in allocs:
            ref = np.fromfile(f"{prefix}.0.a{a}", dtype=np.uint16)
            for i in range(1, n_sessions):
                cur = np.fromfile(f"{prefix}.{i}.a{a}", dtype=np.uint16)
                bad = np.flatnonzero(ref != cur)
                msg = f"DUMP allocation {a} ({NAMES[a]}) session {i} vs 0: {bad.size} of {ref.size} 16-bit words differ"
                if bad.size and NAMES[a] == "qrot":
                    HD = cfg.dim
                    rows, cols = bad // HD, bad % HD
                    msg += f"; rows {sorted(set(rows.tolist()))[:40]}; heads {sorted(set((cols // 128).tolist()))[:40]}; dims {sorted(set((cols % 128).tolist()))[:64]}"
                    msg += f"; first: " + ", ".join(f"[{r},{c}] {ref[k]:04x}->{cur[k]:04x}" for r, c, k in list(zip(rows, cols, bad))[:8])
                elif bad.size:
                    msg += f"; first indices {bad[:16].tolist()}"
                print(msg)

    def load(i):
        d = {}
        for ln in Path(f"{prefix}.{i}").read_text().splitlines():
            f = ln.split()
            d[(int(f[0]), int(f[1]), int(f[3]))] = (f[2], f[4], f[5])
        return d
    ref = load(0)
    print(f"TRACE session 0: {len(ref)} (step, op, allocation) records")
    for i in range(1, n_sessions):
        t = load(i)
        keys = sorted(set(ref) | set(t))
        diffs = [k for k in keys if ref.get(k, (None, None, None))[2] != t.get(k, (None, None, None))[2]]
        if not diffs:
            print(f"TRACE session {i}: identical to session 0")
            continue
        print(f"TRACE session {i}: {len(diffs)} records differ; the first 10:")
        for k in diffs[:10]:
            a, b = ref.get(k), t.get(k)
            site = (a or b)[0]
            name = NAMES[k[2]] if k[2] < len(NAMES) else "?"
            print(f"    step {k[0]} op {k[1]} site {site} allocation {k[2]} ({name}, {(a or b)[1]} bytes): {a[2] if a else 'unchanged'} vs {b[2] if b else 'unchanged'}")


if __name__ == "__main__":
    main()
