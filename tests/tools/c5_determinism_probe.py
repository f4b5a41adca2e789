"""C5 at 64 sessions on a real MI355X: is it the ENGINE or the CHECKER that moves between runs?  

    python tests/tools/c5_determinism_probe.py <tag> [--quant none]

One process = one fresh engine handle + one fresh oracle.  Prints, per step, a hash of everything the engine produced (ring
tokens, text logits, audio logits, the residual stream after the first and the last temporal layer) and of everything the oracle
produced, then repeats the same two steps on a NEW stream of the same handle three more times (same inputs, same forced tokens)
and says whether every repeat reproduced the first run's bits.  Comparing the printed hashes of several processes separates
"engine differs between processes", "oracle differs between processes" and "engine differs between two streams of one process".
Also prints, per sampling site, the share of (row, site) pairs whose logits are bit-identical to the oracle's.
TEST TOOL: uses oracle/ as the checker."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from moshi_amd.config import LMConfig          # noqa: E402
from moshi_amd.weights import quantize_lm_state_dict, random_lm_state_dict      # noqa: E402
from oracle.lm_oracle import LMOracle           # noqa: E402
from tests import lm_cases                       # noqa: E402


def h(*arrays):
    m = hashlib.sha1()
    for a in arrays:
        m.update(np.ascontiguousarray(a).tobytes())
    return m.hexdigest()[:12]


def main():
    tag = sys.argv[1]
    quant = "q8" if "--quant" not in sys.argv else sys.argv[sys.argv.index("--quant") + 1]
    B, S, seed = 64, 2, 364
    cfg = LMConfig(num_layers=2, context=64)
    sd = random_lm_state_dict(cfg, seed=seed)
    if quant == "q8":
        sd = quantize_lm_state_dict(sd)
    gen = lm_cases.make_engine(cfg, sd, "cuda", None, B, use_sampling=False, support_out_of_sync=True)
    gen.lm_model.enable_hidden_taps()
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
        plan.append(dict(mask=mask, reset=reset, codes=codes, forced=np.concatenate([ott[:, None], oat], 1), otl=otl, oal=oal, oo=oo,
                         otap=[orc.hidden_taps[0].copy(), orc.hidden_taps[1].copy()]))
        print(f"ORACLE {tag} step {s}: tokens {h(oo)} text {h(otl)} audio {h(oal)} tap0 {h(orc.hidden_taps[0])} tap1 {h(orc.hidden_taps[1])}")

    def run_engine():
        out = []
        with gen.streaming(B):
            for p in plan:
                if p["reset"] is not None:
                    gen.reset_streaming(torch.from_numpy(p["reset"]).to("cuda"))
                gen.set_exec_mask(torch.from_numpy(p["mask"]).to("cuda"))
                o, tl, al = gen.step_with_taps(torch.from_numpy(p["codes"]).to("cuda"), forced_tokens=torch.from_numpy(p["forced"]).to("cuda"))
                taps = gen.hidden_taps().float().cpu().numpy()
                out.append(dict(o=o.cpu().numpy(), tl=tl.cpu().numpy(), al=al.cpu().numpy(), tap=taps))
        return out
    first = run_engine()
    for s, r in enumerate(first):
        m = plan[s]["mask"]
        print(f"ENGINE {tag} step {s}: tokens {h(r['o'][m])} text {h(r['tl'][m])} audio {h(r['al'][m])} tap0 {h(r['tap'][0][m])} tap1 {h(r['tap'][1][m])}")
    for rep in range(3):
        again = run_engine()
        diffs = []
        for s, (a, b) in enumerate(zip(first, again)):
            m = plan[s]["mask"]
            for key in ("o", "tl", "al"):
                if not np.array_equal(a[key][m], b[key][m]):
                    diffs.append(f"step{s}.{key}")
            for w in (0, 1):
                if not np.array_equal(a["tap"][w][m], b["tap"][w][m]):
                    rows = np.flatnonzero((a["tap"][w][m] != b["tap"][w][m]).any(1))
                    diffs.append(f"step{s}.tap{w}(rows {rows[:6].tolist()}..{len(rows)})")
        print(f"REPEAT {tag} {rep}: " + ("bit-identical to the first run" if not diffs else "DIFFERS: " + " ".join(diffs)))
    # where do engine and oracle part ways: per sampling site, and at the taps
    for s, r in enumerate(first):
        m = plan[s]["mask"]
        same_text = float(np.mean([(r["tl"][b] == plan[s]["otl"][b]).all() for b in np.flatnonzero(m)]))
        same_audio = [float(np.mean([(r["al"][b, k] == plan[s]["oal"][b, k]).all() for b in np.flatnonzero(m)])) for k in range(cfg.dep_q)]
        tap_same = [float(np.mean([(r["tap"][w][b] == plan[s]["otap"][w][b]).all() for b in np.flatnonzero(m)])) for w in (0, 1)]
        print(f"VS_ORACLE {tag} step {s}: rows bit-identical - tap0 {tap_same[0]:.3f} tap1 {tap_same[1]:.3f} text {same_text:.3f} audio " +
              " ".join(f"{v:.3f}" for v in same_audio))


if __name__ == "__main__":
    main()
