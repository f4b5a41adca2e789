"""Parity cases for LMGen.step shared by the simulator tests (CPU) and the GPU tests."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from moshi_amd.config import LMConfig, tiny_lm_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.weights import random_lm_state_dict
from oracle.lm_oracle import LMOracle

GOLDEN = Path(__file__).resolve().parent / "golden"

# Logits tolerance (bf16 model, fp32 accumulation; only summation order / attention-backend rounding differ):
#   max |d| <= LOGIT_MAX_REL * max|ref|   and   mean |d| <= LOGIT_MEAN_REL * max|ref|   per (row, sampling site).
# Yardstick: the distance between two CORRECT implementations of this bf16 model on the same inputs - the reference's PyTorch
# CPU path (the golden vectors) and the numpy oracle - is max 3.12 % / mean 0.90 % of max|logit| in the worst of 162 (row,
# site) pairs (median 1.2 % / 0.4 %): pure bf16 rounding-order noise through 2 + 8 x 2 transformer layers (the reviewer of
# round 4 measured the reference moving 1.2 % against ITSELF between two hosts).  The gate is that distance x 1.6 (max) and
# x 1.33 (mean): above the yardstick, because an engine is a third correct implementation and cannot be asked to sit closer to
# the reference than the checker does, and below twice the yardstick - tests/test_oracle_pinned.py
# (test_bf16_logit_gate_is_under_twice_the_distance_between_two_correct_implementations) re-measures the yardstick and holds
# the gate inside [1 x, 2 x).  The ENGINE against the same goldens on MI355X sits AT the yardstick
# (profiles/r05_logs/parity_golden_{tiny,wide}_cuda.json: tiny worst max 3.12 % / mean 0.90 %, 7B widths x 1 layer worst max
# 2.33 % / mean 0.51 %).
LOGIT_MAX_REL, LOGIT_MEAN_REL = 0.05, 0.012


def logits_close(a: np.ndarray, ref: np.ndarray, widen: float = 1.0) -> bool:
    scale = float(np.abs(ref).max()) + 1e-6
    d = np.abs(a - ref)
    return float(d.max()) <= widen * LOGIT_MAX_REL * scale and float(d.mean()) <= widen * LOGIT_MEAN_REL * scale


# Guided logits are null + (cond - null) * coef: the rounding noise of BOTH model rows enters, weighted |coef| and |coef - 1|,
# while the scale of the result grows less than that.  Measured on tests/golden/lm_cfg.npz: the numpy oracle sits at max 4.0 % /
# mean 1.4 % of max|ref| from the reference at coef 2 (3.2 % / 0.8 % at coef 3 with a condition) -> tolerance widened 1.5x.
GUIDED_WIDEN = 1.5


def near_tie(ref_logits: np.ndarray, tok_a: int, tok_b: int, widen: float = 1.0) -> bool:
    """Two implementations may pick different tokens only where the reference logits nearly tie."""
    scale = float(np.abs(ref_logits).max()) + 1e-6
    return abs(float(ref_logits[tok_a]) - float(ref_logits[tok_b])) <= 2 * widen * LOGIT_MAX_REL * scale


_SD_CACHE: "dict" = {}


def cached_lm_state_dict(cfg, seed: int, keep: int = 2):
    """`random_lm_state_dict(cfg, seed)` (the CPU draw the golden generators and every earlier measurement used), kept for the
    next test that asks for the same (architecture, seed): drawing ~1 B parameters on the host is 10 s of each full-width test,
    and the GPU suite has a time budget.  The tensors are shared: callers must not modify them."""
    key = (repr(cfg), int(seed))
    if key not in _SD_CACHE:
        while len(_SD_CACHE) >= keep:
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
        _SD_CACHE[key] = random_lm_state_dict(cfg, seed=seed)
    else:
        _SD_CACHE[key] = _SD_CACHE.pop(key)          # most recently used last
    return dict(_SD_CACHE[key])


def make_engine(cfg, sd, device, lib, max_batch, **gen_kwargs):
    lm = LMModel(sd, cfg, device=device, max_batch=max_batch, lib=lib)
    return LMGen(lm, **gen_kwargs)


def check_golden_greedy(device, lib):
    """Replays the reference's greedy run (exec masks + partial reset), teacher-forced with the reference's own
    tokens so that a near-tie cannot cascade: ring outputs must be identical, logits within tolerance."""
    g = np.load(GOLDEN / "lm_tiny.npz")
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))
    gen = make_engine(cfg, sd, device, lib, 3, use_sampling=False, support_out_of_sync=True)
    S, B = g["masks"].shape
    agree = total = 0
    log = ErrorLog()
    with gen.streaming(B):
        for s in range(S):
            if s == int(g["reset_step"][0]):
                gen.reset_streaming(torch.from_numpy(g["reset_mask"]).to(device))
            gen.set_exec_mask(torch.from_numpy(g["masks"][s]).to(device))
            forced = np.concatenate([g["g_text_tok"][s][:, None], g["g_audio_tok"][s]], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(g["codes"][s]).to(device),
                                             forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            for b in range(B):
                if not g["masks"][s, b]:
                    assert (out[b] == -2).all()            # lm.py:781-782
                    continue
                assert np.array_equal(out[b], g["g_tokens"][s, b]), f"step {s} row {b}: ring output differs"
                log.add("text", tl[b], g["g_text_logits"][s, b])
                assert logits_close(tl[b], g["g_text_logits"][s, b]), f"step {s} row {b}: text logits"
                t_eng, t_ref = int(tl[b].argmax()), int(g["g_text_tok"][s, b])
                assert t_eng == t_ref or near_tie(g["g_text_logits"][s, b], t_eng, t_ref)
                for k in range(cfg.dep_q):
                    log.add(f"audio{k}", al[b, k], g["g_audio_logits"][s, b, k])
                    assert logits_close(al[b, k], g["g_audio_logits"][s, b, k]), f"step {s} row {b} cb {k}: audio logits"
                    a_eng, a_ref = int(al[b, k].argmax()), int(g["g_audio_tok"][s, b, k])
                    assert a_eng == a_ref or near_tie(g["g_audio_logits"][s, b, k], a_eng, a_ref)
                    agree += a_eng == a_ref
                    total += 1
    log.dump(f"golden_tiny_{torch.device(device).type}")
    assert agree >= 0.85 * total, f"greedy agreement with the reference too low: {agree}/{total}"


def load_wide():
    """Reference run at Moshi-7B's real layer shapes (1 temporal layer; tests/golden/make_golden_lm.py gen_lm_wide)."""
    g = dict(np.load(GOLDEN / "lm_wide.npz"))
    for k in ("g_text_logits", "g_audio_logits"):
        g[k] = (g.pop(k + "_bf16").astype(np.uint32) << 16).view(np.float32)
    return g, LMConfig(num_layers=1, context=16)


def check_wide_steps(step_fn, g, cfg, name=None, widen=1.0, set_mask=None):
    """`step_fn(codes, forced) -> (ring output, text logits, audio logits)` replayed over a golden greedy run of the reference
    (lm_wide.npz / lm_full.npz): ring outputs identical, logits within `widen` x the tolerance, argmax equal or a near-tie of
    the reference's logits.  `name`: the measured errors are printed and written to gpurun_out/parity_<name>.json."""
    S, B = g["g_text_tok"].shape
    log = ErrorLog()
    for s in range(S):
        mask = g["masks"][s] if "masks" in g else np.ones(B, bool)
        if set_mask is not None:
            set_mask(mask)
        forced = np.concatenate([g["g_text_tok"][s][:, None], g["g_audio_tok"][s]], 1)
        out, tl, al = step_fn(g["codes"][s], forced)
        for b in range(B):
            if not mask[b]:
                assert (out[b] == -2).all()            # lm.py:781-782
                continue
            assert np.array_equal(out[b], g["g_tokens"][s, b]), f"step {s} row {b}: ring output differs"
            log.add("text", tl[b], g["g_text_logits"][s, b])
            assert logits_close(tl[b], g["g_text_logits"][s, b], widen), f"step {s} row {b}: text logits"
            t_e, t_r = int(tl[b].argmax()), int(g["g_text_tok"][s, b])
            assert t_e == t_r or near_tie(g["g_text_logits"][s, b], t_e, t_r, widen)
            for k in range(cfg.dep_q):
                log.add(f"audio{k}", al[b, k], g["g_audio_logits"][s, b, k])
                assert logits_close(al[b, k], g["g_audio_logits"][s, b, k], widen), f"step {s} row {b} cb {k}: audio logits"
                a_e, a_r = int(al[b, k].argmax()), int(g["g_audio_tok"][s, b, k])
                assert a_e == a_r or near_tie(g["g_audio_logits"][s, b, k], a_e, a_r, widen)
    if name:
        log.dump(name)
    return log


def check_golden_wide(device, lib):
    """The engine at the 7B layer shapes against the reference's own output (greedy, teacher-forced)."""
    g, cfg = load_wide()
    sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))     # drawn on the CPU, like the generator script: the CUDA RNG stream differs
    B = g["codes"].shape[1]
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    with gen.streaming(B):
        def step(codes, forced):
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            return out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
        check_wide_steps(step, g, cfg, name=f"golden_wide_{torch.device(device).type}")


# The reference ITSELF at the benchmark depth (tests/golden/make_golden_lm_full.py): Moshi-7B as loaders._lm_kwargs builds it,
# 32 temporal layers, bf16 on the CPU, B = 2 with the rows one step apart, 4 greedy steps.  Two correct bf16 implementations
# drift apart layer by layer under random-init weights (see FULL_DEPTH_FACTOR below), so the logits gate is FULL_WIDEN x the
# shallow-model tolerance.  Yardstick: the distance between the reference and the numpy oracle at 32 layers - two correct
# implementations - is worst max 6.10 % / mean 1.15 % of max|logit| (parity_golden_full_oracle); gate = yardstick x 1.5 = 9.2 %
# -> FULL_WIDEN = 9.2 / 5 (mean: 1.2 % x 1.85 = 2.2 %, x 1.9 the yardstick's mean); tests/test_oracle_pinned.py re-measures the
# yardstick and holds the gate inside [1 x, 2 x) of it.  The engine on MI355X against the same file: worst max 4.85 % / mean
# 1.09 % (text head 2.30 % / 0.44 %; profiles/r05_logs/parity_golden_full_cuda.json; round 3: 5.77 % / 1.13 %) - inside the yardstick.
FULL_WIDEN = 1.85


def load_full():
    g = dict(np.load(GOLDEN / "lm_full.npz"))
    for k in ("g_text_logits", "g_audio_logits"):
        g[k] = (g.pop(k + "_bf16").astype(np.uint32) << 16).view(np.float32)
    return g, LMConfig()


_FULL_SD = {}


def full_golden_state_dict(device):
    """The 7.7 B parameters of tests/golden/lm_full.npz's model: drawn ONCE per process on the host (the generator's draw - the
    CUDA RNG stream differs) and kept on the device for the three tests that read them (25 s of host RNG each otherwise)."""
    g, cfg = load_full()
    key = (int(g["seed"][0]), str(device))
    if key not in _FULL_SD:
        _FULL_SD.clear()
        sd = random_lm_state_dict(cfg, seed=key[0])
        if torch.device(device).type == "cuda":
            sd = {k: v.to(device) for k, v in sd.items()}
        _FULL_SD[key] = sd
    return dict(_FULL_SD[key])


def release_full_golden_state_dict():
    _FULL_SD.clear()


def check_golden_full(device, lib, max_batch=None, name=None):
    """The engine on the benchmark model (32 layers, context 3000) against the reference's own output, teacher-forced.
    max_batch: build the handle for that many sessions (> 16: the 32-row MFMA tile and k_gemm_xlds - the benchmark's kernels -
    instead of the 16-row tile a 2-session handle gets)."""
    g, cfg = load_full()
    sd = full_golden_state_dict(device)
    B = g["codes"].shape[1]
    gen = make_engine(cfg, sd, device, lib, max_batch or B, use_sampling=False, support_out_of_sync=True)
    del sd
    with gen.streaming(B):
        def step(codes, forced):
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            return out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
        return check_wide_steps(step, g, cfg, name=name or f"golden_full_{torch.device(device).type}", widen=FULL_WIDEN,
                                set_mask=lambda m: gen.set_exec_mask(torch.from_numpy(m).to(device)))


def check_free_running(device, lib, g, cfg, widen, name, max_batch=None, on_cuda_weights=False, sd=None):
    """NOT teacher-forced: the engine runs greedy on its own tokens through a golden run of the reference - what
    it samples at step s is what it reads at step s + 1.  Per row, the first (step, site) where its token differs from the
    reference's is reported, and it must be a near-tie of the REFERENCE's logits at that site (all inputs up to there were
    identical, so those logits are comparable); from there on the row is on another trajectory and is not compared (until a
    reset of that row puts it back on the reference's)."""
    import json
    if sd is None:
        sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))
        if on_cuda_weights and torch.device(device).type == "cuda":
            sd = {k: v.to(device) for k, v in sd.items()}
    S, B = g["g_text_tok"].shape
    gen = make_engine(cfg, sd, device, lib, max_batch or B, use_sampling=False, support_out_of_sync=True)
    del sd
    off_track, events = {}, []
    compared = 0
    masks = g["masks"] if "masks" in g else np.ones((S, B), bool)
    reset_step = int(g["reset_step"][0]) if "reset_step" in g else -1
    with gen.streaming(B):
        for s in range(S):
            if s == reset_step:
                gen.reset_streaming(torch.from_numpy(g["reset_mask"]).to(device))
                for b in np.flatnonzero(g["reset_mask"]):
                    off_track.pop(int(b), None)
            gen.set_exec_mask(torch.from_numpy(masks[s]).to(device))
            out, tl, al = gen.step_with_taps(torch.from_numpy(g["codes"][s]).to(device))
            tl, al = tl.cpu().numpy(), al.cpu().numpy()
            for b in range(B):
                if not masks[s, b] or b in off_track:
                    continue
                sites = [("text", tl[b], g["g_text_logits"][s, b], int(g["g_text_tok"][s, b]))]
                sites += [(f"audio{k}", al[b, k], g["g_audio_logits"][s, b, k], int(g["g_audio_tok"][s, b, k])) for k in range(cfg.dep_q)]
                for site, lg, ref, tok_ref in sites:
                    tok = int(lg.argmax())
                    compared += 1
                    if tok != tok_ref:
                        scale = float(np.abs(ref).max()) + 1e-6
                        off_track[b] = True
                        events.append({"row": b, "step": s, "site": site, "engine": tok, "reference": tok_ref,
                                       "reference_logit_gap_rel": abs(float(ref[tok]) - float(ref[tok_ref])) / scale,
                                       "near_tie": bool(near_tie(ref, tok, tok_ref, widen))})
                        break               # later sites of this step already condition on the different token
    res = {"case": name, "steps": int(S), "rows": int(B), "token_decisions_compared": compared, "divergences": events}
    print(f"[parity] {name}: free-running greedy vs the reference, {compared} token decisions compared on the reference's trajectory; "
          + ("no row left it" if not events else "; ".join(f"row {v['row']}: step {v['step']} {v['site']} (reference logit gap "
                                                          f"{v['reference_logit_gap_rel']:.4f} of max|logit|, near-tie: {v['near_tie']})" for v in events)))
    out = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out.is_dir():
        (out / f"parity_{name}.json").write_text(json.dumps(res, indent=1))
    for v in events:
        assert v["near_tie"], f"row {v['row']} left the reference's trajectory at step {v['step']} ({v['site']}) where its logits do not tie: {v}"
    return res


def check_golden_full_free_running(device, lib, max_batch=None, name="golden_full_free_running"):
    """The benchmark model (32 layers, context 3000) free-running against the reference's own run (lm_full.npz)."""
    g, cfg = load_full()
    return check_free_running(device, lib, g, cfg, FULL_WIDEN, name, max_batch=max_batch, sd=full_golden_state_dict(device))


def check_golden_tiny_free_running(device, lib):
    """The same on the tiny golden schedule (exec masks + a partial reset, which puts a row back on the reference's trajectory)."""
    return check_free_running(device, lib, np.load(GOLDEN / "lm_tiny.npz"), tiny_lm_config(), 1.0,
                              f"golden_tiny_free_running_{torch.device(device).type}", max_batch=3)


def check_golden_sampled(device, lib):
    """Replays the reference's sampled run with the Exp(1) draws recorded at its `multinomial`."""
    g = np.load(GOLDEN / "lm_tiny.npz")
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))
    gen = make_engine(cfg, sd, device, lib, 3, use_sampling=True, temp=0.8, temp_text=0.7, top_k=20, top_k_text=10,
                      support_out_of_sync=True)
    S, B = g["s_text_tok"].shape
    same = total = 0
    with gen.streaming(B):
        for s in range(S):
            forced = np.concatenate([g["s_text_tok"][s][:, None], g["s_audio_tok"][s]], 1)
            # first pass WITHOUT forcing on a scratch copy is not possible (state); instead: force the history and
            # compare the engine's own noisy choice through the oracle-free identity  argmax(p/q)  on its logits taps
            out, tl, al = gen.step_with_taps(torch.from_numpy(g["codes2"][s]).to(device),
                                             noise=torch.from_numpy(g["s_noise"][s]).to(device),
                                             forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            assert np.array_equal(out, g["s_tokens"][s])
            from oracle.lm_oracle import sample_token
            tt = sample_token(tl, True, 0.7, 10, g["s_noise"][s][:, 0])
            same += int((tt == g["s_text_tok"][s]).sum()); total += B
            for k in range(cfg.dep_q):
                assert logits_close(al[:, k], g["s_audio_logits"][s][:, k])
                at = sample_token(al[:, k], True, 0.8, 20, g["s_noise"][s][:, 1 + k])
                same += int((at == g["s_audio_tok"][s][:, k]).sum()); total += B
    assert same >= 0.9 * total, f"sampled-token agreement given the reference's noise: {same}/{total}"


def engine_sampling_matches_oracle_rule(device, lib, cfg=None, top_k=20, top_k_text=10, B=4, steps=4):
    """The engine's in-kernel sampler (softmax -> top-k -> argmax(p/q)) against the oracle's restatement of
    sampling.py:86-106 on the engine's OWN logits with supplied noise: must be identical token for token."""
    from oracle.lm_oracle import sample_token
    cfg = cfg or tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=5)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=True, temp=0.8, temp_text=0.7, top_k=top_k, top_k_text=top_k_text,
                      support_out_of_sync=True)
    rng = np.random.default_rng(1)
    kmax = max(top_k, top_k_text)
    with gen.streaming(B):
        for s in range(steps):
            codes = rng.integers(0, cfg.card, (B, 8, 1))
            noise = rng.exponential(1.0, (B, 1 + cfg.dep_q, kmax)).astype(np.float32)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), noise=torch.from_numpy(noise).to(device))
            tl, al = tl.cpu().numpy(), al.cpu().numpy()
            # the step output holds text of this step (delay 0) only after the ring delay; re-derive from taps instead
            tt = sample_token(tl, True, 0.7, top_k_text, noise[:, 0])
            toks = [sample_token(al[:, k], True, 0.8, top_k, noise[:, 1 + k]) for k in range(cfg.dep_q)]
            if s >= 1:
                o = out.cpu().numpy()[:, :, 0]
                # channels with delay 1 (acoustic codebooks 1..) are emitted in the step they are sampled
                for k in range(1, cfg.dep_q):
                    assert np.array_equal(o[:, 1 + k], toks[k]), (s, k)
                assert np.array_equal(o[:, 0], prev_tt) and np.array_equal(o[:, 1], prev_a0)
            prev_tt, prev_a0 = tt, toks[0]


def oracle_vs_engine(device, lib, cfg, seed, B, S, use_masks=True, quantize=False, input_scale=1.0, stats=None, int8_activations=True):
    if quantize is True and int8_activations:      # int8 x int8 linears: per-linear bit equality + a statistical network gate (below)
        return int8_network_vs_oracle(device, lib, cfg, seed, B, S, use_masks=use_masks, stats=stats)
    sd = cached_lm_state_dict(cfg, seed)
    if quantize == "fp8":   # e4m3fn linears on the fp8 MFMA (BASELINE configs[4]); engine and oracle get the same fp8 tensors
        from moshi_amd.weights import quantize_lm_state_dict_fp8
        sd = quantize_lm_state_dict_fp8(sd, input_scale=input_scale)
    elif quantize:  # row-wise int8 linears (the reference's `quantize=True` storage); engine and oracle get the same int8 tensors
        from moshi_amd.weights import quantize_lm_state_dict
        sd = quantize_lm_state_dict(sd)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg, int8_activations=int8_activations)      # int8 linears here: weight-only (the int8 x int8 rule: above)
    widen = 1.0
    orc.streaming(B)
    rng = np.random.default_rng(seed)
    with gen.streaming(B):
        for s in range(S):
            mask = np.ones(B, bool)
            if use_masks and B > 1:
                mask = rng.random(B) > 0.3
                mask[0] = True
                if s == S // 2:
                    r = np.zeros(B, bool); r[B - 1] = True
                    orc.reset_streaming(r); gen.reset_streaming(torch.from_numpy(r).to(device))
                    mask[B - 1] = True
            orc.set_exec_mask(mask); gen.set_exec_mask(torch.from_numpy(mask).to(device))
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            for b in range(B):
                if not mask[b]:
                    continue
                assert np.array_equal(out[b], oo[b]), f"step {s} row {b}: ring output differs"
                assert logits_close(tl[b], otl[b], widen), f"step {s} row {b}: text logits {np.abs(tl[b]-otl[b]).max()}"
                for k in range(cfg.dep_q):
                    assert logits_close(al[b, k], oal[b, k], widen), f"step {s} row {b} cb {k}: {np.abs(al[b,k]-oal[b,k]).max()}"
                    a_e, a_o = int(al[b, k].argmax()), int(oat[b, k])
                    assert a_e == a_o or near_tie(oal[b, k], a_e, a_o, widen)
        if stats is not None:
            import collections
            stats["xlds_launches"] = int(gen._lib.mmi_lm_stat(gen.lm_model._handle, 0))
            stats["launch_sites"] = dict(collections.Counter(site for site, _ in gen.launch_list()))
            stats["launch_list"] = list(gen.launch_list())
            stats["dep_tile"] = int(gen._lib.mmi_lm_stat(gen.lm_model._handle, 3))


class ErrorLog:
    """Measured engine-vs-checker logit errors per sampling site, relative to max|ref| of the row: what the tolerance is set from."""

    def __init__(self):
        self.rows = {}

    def add(self, site: str, a: np.ndarray, ref: np.ndarray):
        scale = float(np.abs(ref).max()) + 1e-6
        d = np.abs(a - ref)
        self.rows.setdefault(site, []).append((float(d.max()) / scale, float(d.mean()) / scale))

    def summary(self) -> dict:
        out = {}
        for site, v in self.rows.items():
            mx, mn = np.array([x[0] for x in v]), np.array([x[1] for x in v])
            out[site] = {"n": len(v), "max_rel_worst": float(mx.max()), "max_rel_median": float(np.median(mx)),
                         "mean_rel_worst": float(mn.max()), "mean_rel_median": float(np.median(mn))}
        return out

    def dump(self, name: str):
        import json
        summ = self.summary()
        worst = max(v["max_rel_worst"] for v in summ.values())
        worst_mean = max(v["mean_rel_worst"] for v in summ.values())
        print(f"[parity] {name}: worst max-rel {worst:.4f}, worst mean-rel {worst_mean:.4f} over {sum(v['n'] for v in summ.values())} (row, site) pairs")
        for site, v in summ.items():
            print(f"[parity]   {site:8s} max-rel worst {v['max_rel_worst']:.4f} median {v['max_rel_median']:.4f} | "
                  f"mean-rel worst {v['mean_rel_worst']:.4f} median {v['mean_rel_median']:.4f}")
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        if out.is_dir():
            (out / f"parity_{name}.json").write_text(json.dumps({"case": name, "sites": summ}, indent=1))
        return worst, worst_mean


class HiddenLog:
    """The residual stream after the first / the last temporal layer, engine vs checker, in units of one bf16 ulp at the row's
    RMS (2^-7 * rms(row): the spacing of bf16 values of typical magnitude in that row), per element.  The logit gate (5 % of
    max|logit|) is a wide net at the END of the network; this is the net in the middle: after ONE layer two correct bf16
    implementations differ by isolated last-bit flips, after all of them by the accumulated drift the yardstick measures."""

    def __init__(self):
        self.units = {0: [], 1: []}

    def add(self, which: int, a: np.ndarray, ref: np.ndarray):
        rms = np.sqrt((ref.astype(np.float64) ** 2).mean(-1, keepdims=True)) + 1e-30
        self.units[which].append((np.abs(a.astype(np.float64) - ref) / (rms * 2.0 ** -7)).ravel())

    def summary(self) -> dict:
        out = {}
        for w, v in self.units.items():
            if not v:
                continue
            u = np.concatenate(v)
            out[w] = {"n": int(u.size), "mean": float(u.mean()), "p99": float(np.quantile(u, 0.99)), "max": float(u.max()),
                      "exact": float((u == 0).mean())}
        return out

    def check_against(self, yard: "HiddenLog", name: str, num_layers: int):
        import json
        es, ys = self.summary(), yard.summary()
        for w, label in ((0, "layer 0"), (1, f"layer {num_layers - 1}")):
            print(f"[parity] {name} after {label}: engine mean {es[w]['mean']:.3f} p99 {es[w]['p99']:.2f} max {es[w]['max']:.1f} bf16 ulps at the "
                  f"row RMS, {100 * es[w]['exact']:.1f} % bit-equal | fp64-vs-fp32 oracle mean {ys[w]['mean']:.3f} p99 {ys[w]['p99']:.2f} max {ys[w]['max']:.1f}")
        out = Path(__file__).resolve().parent.parent / "gpurun_out"
        if out.is_dir():
            (out / f"parity_{name}.json").write_text(json.dumps({"case": name, "unit": "bf16 ulp at the row RMS", "engine": es, "yardstick": ys}, indent=1))
        # after ONE layer: isolated last-bit flips only - the mean is a small fraction of an ulp whatever the summation order
        assert es[0]["mean"] <= max(0.25, FULL_DEPTH_FACTOR * ys[0]["mean"]), f"layer 0: {es[0]} vs yardstick {ys[0]}"
        assert es[0]["p99"] <= max(2.0, FULL_DEPTH_FACTOR * ys[0]["p99"]), f"layer 0: {es[0]} vs yardstick {ys[0]}"
        # after the last layer: within FULL_DEPTH_FACTOR x what a different (fp64) accumulation order alone produces
        for key in ("mean", "p99"):
            assert es[1][key] <= FULL_DEPTH_FACTOR * ys[1][key] + 0.25, f"last layer {key}: engine {es[1][key]:.3f} vs yardstick {ys[1][key]:.3f}"


def lazy_temporal_linears(sd):
    """The checker keeps the temporal transformer's big linears where they are (bf16 on the GPU) and widens each to fp32 only
    while it multiplies with it (oracle.lm_oracle.LazyWeight): the 32-layer oracle then needs ~2 GB of host memory, not 30."""
    from oracle.lm_oracle import LazyWeight
    out = {}
    for k, v in sd.items():
        big = k.startswith("transformer.layers.") and k.endswith(".weight")
        out[k] = LazyWeight(v) if big else v
    return out


# Full depth (32 temporal layers, random-init weights): a bf16 rounding flip in one GEMM output is amplified layer by layer, so
# two CORRECT implementations that differ only in summation order drift apart by several % of max|logit|.  The yardstick is
# measured inside the test: the oracle accumulating every GEMM in fp64 against the oracle accumulating in fp32 (same weights,
# same inputs, same forced tokens).  The engine must sit within FULL_DEPTH_FACTOR x that distance from the fp32 oracle, per
# sampling site, for the worst and the median row (first GPU run, profiles/r02_logs/parity_full_depth_b32_first_run.json:
# engine 4.0 % median / 6.0 % worst max-rel on the text logits, 5-8 % / 8-13.5 % on the audio sites).
FULL_DEPTH_FACTOR = 1.5


def full_depth_vs_oracle(device, lib, B=32, S=3, num_layers=32, seed=4242, name="full_depth_b32", cfg=None):
    """The model bench.py times - LMConfig(): 32 temporal layers, context 3000, B sessions - against the numpy oracle,
    teacher-forced, rows at different depths.  The checker's ring is shortened to 64 slots (no wrap inside this test; the
    real 3000-slot wrap is `ring_wrap_at_real_capacity`), the engine runs its real ring."""
    from dataclasses import replace
    cfg = cfg or LMConfig(num_layers=num_layers)
    sd = random_lm_state_dict(cfg, seed=seed, device=device)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    gen.lm_model.enable_hidden_taps()              # the residual stream after temporal layers 0 and num_layers - 1
    hid, hid_yard = HiddenLog(), HiddenLog()
    lazy = lazy_temporal_linears(sd)
    orc = LMOracle(lazy, replace(cfg, context=64))
    ref64 = LMOracle(lazy, replace(cfg, context=64), accumulate64=True)     # the yardstick's second implementation
    orc.streaming(B); ref64.streaming(B)
    rng = np.random.default_rng(seed)
    log, yard = ErrorLog(), ErrorLog()
    start = (np.arange(B) % 5) * 7                 # rows at different stream positions (0 .. 28), none near the 64-slot checker ring
    with gen.streaming(B):
        gen.seek(start); orc.seek(start); ref64.seek(start)
        for s in range(S):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            _, (ytl, yal, _, _) = ref64.step(codes, use_sampling=False, support_out_of_sync=True, forced=forced)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            assert np.array_equal(out, oo), f"step {s}: ring output differs"
            taps = gen.hidden_taps().float().cpu().numpy()
            for w in (0, 1):
                hid.add(w, taps[w], orc.hidden_taps[w]); hid_yard.add(w, ref64.hidden_taps[w], orc.hidden_taps[w])
            for b in range(B):
                log.add("text", tl[b], otl[b]); yard.add("text", ytl[b], otl[b])
                for k in range(cfg.dep_q):
                    log.add(f"audio{k}", al[b, k], oal[b, k]); yard.add(f"audio{k}", yal[b, k], oal[b, k])
    log.dump(name)
    yard.dump(name + "_yardstick_fp64_accumulation")
    hid.check_against(hid_yard, name + "_hidden", cfg.num_layers)
    es, ys = log.summary(), yard.summary()
    bad = []
    for site in es:
        for key in ("max_rel_worst", "max_rel_median", "mean_rel_worst", "mean_rel_median"):
            if es[site][key] > FULL_DEPTH_FACTOR * ys[site][key] + 1e-3:
                bad.append(f"{site}.{key}: engine {es[site][key]:.4f} vs yardstick {ys[site][key]:.4f}")
    assert not bad, "engine further from the fp32 oracle than summation order explains: " + "; ".join(bad[:6])
    return log


def ring_wrap_at_real_capacity(device, lib, B=2, S=14, seed=91):
    """The temporal ring at its real geometry (3000 slots x 32 heads x 128) wrapping: sessions are moved to positions
    2995 / 2990 of an all-zero ring (mmi_lm_seek), then stepped across slot 2999 -> 0; engine and checker hold the same ring."""
    cfg = LMConfig(num_layers=2)
    sd = random_lm_state_dict(cfg, seed=seed, device=device)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg)
    orc.streaming(B)
    rng = np.random.default_rng(seed)
    start = np.array([cfg.context - 5 - 5 * b for b in range(B)])
    log = ErrorLog()
    with gen.streaming(B):
        gen.seek(start); orc.seek(start)
        for s in range(S):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            assert np.array_equal(out, oo), f"step {s}: ring output differs"
            for b in range(B):
                log.add("text", tl[b], otl[b])
                assert logits_close(tl[b], otl[b]), f"step {s} row {b} (position {start[b] + s}): text logits"
                for k in range(cfg.dep_q):
                    assert logits_close(al[b, k], oal[b, k]), f"step {s} row {b} cb {k}"
    log.dump("ring_wrap_3000")


def check_step_hooks(device, lib):
    """LMGen(on_text_logits_hook, on_text_hook, on_audio_hook) (lm.py:568-570, 734-757): the hooks receive the reference's
    tensors ([B,1,1,card] bf16 logits, [B] text tokens, [B,dep_q] audio tokens) and what they write in place is what the step
    goes on with.  A hooked run - logits hook leaving row 1 one finite entry, text hook overwriting row 0's token, audio hook
    overwriting the last codebook - must produce exactly the ring outputs of an un-hooked run that is teacher-forced with
    those tokens (the forced text token reaches the depth transformer in both), and the logits hook must see the un-hooked
    run's logits."""
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=21)
    B, S = 2, 5
    rng = np.random.default_rng(5)
    codes = [torch.from_numpy(rng.integers(0, cfg.card, (B, 8, 1))).to(device) for _ in range(S)]
    seen = {"logits": [], "text": [], "audio": []}

    def on_logits(lg):
        assert lg.shape == (B, 1, 1, cfg.text_card) and lg.dtype == torch.bfloat16
        seen["logits"].append(lg.float().cpu().clone())
        lg[1, 0, 0, :] = -float("inf")
        lg[1, 0, 0, 7] = 0.0                      # row 1 can only pick token 7

    def on_text(tok):
        assert tok.shape == (B,) and tok.dtype == torch.int64
        seen["text"].append(tok.cpu().clone())
        tok[0] = 5                                # row 0's text token is overwritten after sampling

    def on_audio(tok):
        assert tok.shape == (B, cfg.dep_q) and tok.dtype == torch.int64
        seen["audio"].append(tok.cpu().clone())
        tok[:, cfg.dep_q - 1] = 3                 # the last codebook is overwritten before it enters the ring

    lm = LMModel(sd, cfg, device=device, max_batch=B, lib=lib)
    hooked = LMGen(lm, use_sampling=False, support_out_of_sync=True, on_text_logits_hook=on_logits, on_text_hook=on_text,
                   on_audio_hook=on_audio)
    outs_h = []
    with hooked.streaming(B):
        for s in range(S):
            outs_h.append(hooked.step(codes[s]).cpu().numpy())
    plain = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    forced = torch.full((B, 1 + cfg.dep_q), -1, dtype=torch.long)
    forced[0, 0], forced[1, 0], forced[:, cfg.dep_q] = 5, 7, 3
    with plain.streaming(B):
        for s in range(S):
            out, tl, al = plain.step_with_taps(codes[s], forced_tokens=forced.to(device))
            assert np.array_equal(out.cpu().numpy(), outs_h[s]), f"step {s}: hooked run differs from the teacher-forced run"
            assert torch.equal(seen["logits"][s][:, 0, 0], tl.cpu()), f"step {s}: the logits hook saw other logits"
            assert int(seen["text"][s][1]) == 7                                   # the logits hook steered row 1's sampler
            assert int(seen["text"][s][0]) == int(tl[0].argmax())                 # row 0: the model's own greedy token
            for k in range(cfg.dep_q - 1):
                assert int(seen["audio"][s][0, k]) == int(al[0, k].argmax()) and int(seen["audio"][s][1, k]) == int(al[1, k].argmax())
    assert len(seen["logits"]) == len(seen["text"]) == len(seen["audio"]) == S


def rng_sampling_statistics(device, lib, iters=40, tol=0.04):
    """On-device RNG path (no supplied noise; the sampler's production form: top-k by radix select, then the largest
    logit / temp - log(Exp(1)) of the set): token frequencies follow softmax(logits / temp) restricted to the top-k - the
    reference's own self-test of its sampler is a frequency check too (sampling.py:109-127)."""
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=5)
    B = 64
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=True, temp=1.0, temp_text=1.0, top_k=8, top_k_text=8,
                      support_out_of_sync=True, seed=123)
    codes = torch.zeros(B, 8, 1, dtype=torch.long, device=device)
    counts = np.zeros(cfg.text_card)
    with gen.streaming(B):
        for it in range(iters):
            gen.reset_streaming()                      # every step is the first step: identical logits for all rows
            out, tl, al = gen.step_with_taps(codes)
            gen.set_exec_mask(torch.ones(B, dtype=torch.bool, device=device))
            o2, _, _ = gen.step_with_taps(codes)       # the text token of step 0 is emitted at step 1 (delay ring)
            for t in o2[:, 0, 0].cpu().numpy():
                counts[t] += 1
    p = torch.softmax(tl[0].double(), -1).cpu().numpy()
    top = np.argsort(-p)[:8]
    expect = np.zeros_like(p); expect[top] = p[top] / p[top].sum()
    freq = counts / counts.sum()
    assert counts.sum() == iters * B and counts[[i for i in range(len(p)) if i not in top]].sum() == 0
    assert np.abs(freq - expect).max() < tol, (freq[top], expect[top])


def smoke_lm(dev):
    """One tiny LMGen.step on the GPU against the oracle (called by __graft_entry__.smoke)."""
    cfg = tiny_lm_config()
    oracle_vs_engine(dev, None, cfg, seed=3, B=2, S=3, use_masks=False)
    print("smoke: LMGen.step ring outputs identical to the oracle, logits within tolerance")


# ---- classifier-free guidance, sum-conditioning, extra heads (tests/golden/lm_cfg.npz) -------------------
CFG_SCENARIOS = {
    "a": dict(cfg_coef=2.0, masked_until=True),
    "b": dict(cfg_coef=1.5, cfg_is_no_text=True),
    "c": dict(cfg_coef=3.0, cond="cond2"),
    "d": dict(cfg_coef=1.0, cond="cond1", heads=True),
}


def load_cfg_golden():
    g = np.load(GOLDEN / "lm_cfg.npz")
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))
    for i, w in enumerate(g["heads_w"]):
        sd[f"extra_heads.{i}.weight"] = torch.from_numpy(w).to(torch.bfloat16)
    return g, cfg, sd


def check_cfg_scenario(g, cfg, name, start, step, heads=None):
    """Replays scenario `name` of the reference's guided / conditioned runs, teacher-forced with the reference's tokens.
    start(**opts) opens the stream; step(codes, forced, mask, reset_mask_or_None) -> (out [B, 1+dep_q, 1], text logits
    [B, V], audio logits [B, dep_q, card]) - the logits the tokens were sampled from, i.e. after the guidance mix;
    heads() -> [rows >= B, n_heads, dim] extra-head probabilities of the last step."""
    sc = CFG_SCENARIOS[name]
    S, B = g["masks"].shape
    wd = GUIDED_WIDEN if sc["cfg_coef"] != 1.0 else 1.0
    start(cfg_coef=sc["cfg_coef"], cfg_is_no_text=sc.get("cfg_is_no_text", False),
          cfg_is_masked_until=[int(v) for v in g["masked_until"]] if sc.get("masked_until") else None,
          condition_sum=g[sc["cond"]][:, 0] if sc.get("cond") else None)
    for s in range(S):
        reset = g["reset_mask"] if s == int(g["reset_step"][0]) else None
        forced = np.concatenate([g[f"{name}_text_tok"][s][:, None], g[f"{name}_audio_tok"][s]], 1)
        out, tl, al = step(g["codes"][s], forced, g["masks"][s], reset)
        for b in range(B):
            if not g["masks"][s, b]:
                continue
            assert np.array_equal(out[b], g[f"{name}_tokens"][s, b]), f"{name} step {s} row {b}: ring output differs"
            assert logits_close(tl[b], g[f"{name}_text_logits"][s, b], wd), f"{name} step {s} row {b}: text logits"
            for k in range(cfg.dep_q):
                assert logits_close(al[b, k], g[f"{name}_audio_logits"][s, b, k], wd), f"{name} step {s} row {b} cb {k}: audio logits"
        if sc.get("heads") and heads is not None:
            pr = heads()
            for b in range(B):
                if g["masks"][s, b]:
                    assert np.abs(pr[b] - g["d_heads"][s, b]).max() <= 0.02, f"extra heads step {s} row {b}"
                    assert np.abs(pr[b].sum(-1) - 1).max() < 0.02


def check_cfg_engine(device, lib, name):
    """Engine vs the reference's guided / conditioned runs (tests/golden/lm_cfg.npz), same checker as the oracle pin."""
    from moshi_amd.lm import ConditionFuser
    g, cfg, sd = load_cfg_golden()
    sc = CFG_SCENARIOS[name]
    if sc.get("heads"):
        from dataclasses import replace
        cfg = replace(cfg, extra_heads_num_heads=int(g["heads_w"].shape[0]), extra_heads_dim=int(g["heads_w"].shape[1]))
    else:
        sd = {k: v for k, v in sd.items() if not k.startswith("extra_heads.")}
    B = g["masks"].shape[1]
    rows = 2 * B if sc["cfg_coef"] != 1.0 else B
    lm = LMModel(sd, cfg, device=device, max_batch=rows, lib=lib, fuser=ConditionFuser({"sum": ["c"]}) if sc.get("cond") else None)
    state = {}

    def start(cfg_coef, cfg_is_no_text, cfg_is_masked_until, condition_sum):
        cond = None
        if condition_sum is not None:
            t = torch.from_numpy(np.asarray(condition_sum)).to(torch.bfloat16)[:, None]        # [rows, 1, dim]
            cond = {"c": (t, torch.ones(t.shape[:2], dtype=torch.bool))}
        gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, cfg_coef=cfg_coef, cfg_is_no_text=cfg_is_no_text,
                    cfg_is_masked_until=cfg_is_masked_until, condition_tensors=cond)
        gen.streaming_forever(B)
        state["gen"] = gen

    def step(codes, forced, mask, reset):
        gen = state["gen"]
        if reset is not None:
            gen.reset_streaming(torch.from_numpy(reset).to(device))
        gen.set_exec_mask(torch.from_numpy(mask).to(device))
        out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
        return out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()

    def heads():
        gen = state["gen"]
        rows_ = int(gen._lib.mmi_lm_model_rows(lm._handle))
        probs = torch.empty(rows_, cfg.extra_heads_num_heads, cfg.extra_heads_dim, device=device, dtype=torch.float32)
        gen._lib.check(gen._lib.mmi_lm_extra_heads(lm._handle, probs.data_ptr(), gen._stream()))
        return probs.cpu().numpy()
    try:
        check_cfg_scenario(g, cfg, name, start, step, heads)
    finally:
        if "gen" in state:
            state["gen"]._stop_streaming()


# ---- cross-attention conditioning (second half; tests/golden/lm_cross.npz) -----------------------------------------
CROSS_SCENARIOS = {
    "e": dict(cfg_coef=1.0, cross=["cross_e"], sum="sum_e"),
    "f": dict(cfg_coef=2.0, cross=["cross_f1", "cross_f2"], pos_emb=0.5),
}


def load_cross_golden():
    from dataclasses import replace
    g = np.load(GOLDEN / "lm_cross.npz")
    cfg = replace(tiny_lm_config(), cross_attention=True)
    return g, cfg, random_lm_state_dict(cfg, seed=int(g["seed"][0]))


def cross_condition_tensors(g, name):
    """The named condition tensors of scenario `name` as the reference's LMGen receives them, and the fuser built for them."""
    from moshi_amd.lm import ConditionFuser
    sc = CROSS_SCENARIOS[name]
    conds, names = {}, []
    for i, key in enumerate(sc["cross"]):
        t = torch.from_numpy(g[key]).to(torch.bfloat16)
        conds[f"x{i}"] = (t, torch.ones(t.shape[:2], dtype=torch.bool))
        names.append(f"x{i}")
    sums = []
    if sc.get("sum"):
        t = torch.from_numpy(g[sc["sum"]]).to(torch.bfloat16)
        conds["s"] = (t, torch.ones(t.shape[:2], dtype=torch.bool))
        sums = ["s"]
    fuser = ConditionFuser({"sum": sums, "cross": names}, cross_attention_pos_emb="pos_emb" in sc,
                           cross_attention_pos_emb_scale=sc.get("pos_emb", 1.0))
    return conds, fuser


def check_cross_scenario(g, cfg, name, start, step):
    """Replays scenario `name` of tests/golden/lm_cross.npz, teacher-forced with the reference's tokens.
    start(cfg_coef, condition_sum [rows, dim] or None, condition_cross [rows, T_c, dim]); step as in check_cfg_scenario."""
    sc = CROSS_SCENARIOS[name]
    S, B = g["masks"].shape
    wd = GUIDED_WIDEN if sc["cfg_coef"] != 1.0 else 1.0
    conds, fuser = cross_condition_tensors(g, name)
    cs, cx = fuser.get_sum(conds), fuser.get_cross(conds)
    start(cfg_coef=sc["cfg_coef"], condition_sum=None if cs is None else cs[:, 0].float().numpy(), condition_cross=cx.float().numpy())
    for s in range(S):
        reset = g["reset_mask"] if s == int(g["reset_step"][0]) else None
        forced = np.concatenate([g[f"{name}_text_tok"][s][:, None], g[f"{name}_audio_tok"][s]], 1)
        out, tl, al = step(g["codes"][s], forced, g["masks"][s], reset)
        for b in range(B):
            if not g["masks"][s, b]:
                continue
            assert np.array_equal(out[b], g[f"{name}_tokens"][s, b]), f"{name} step {s} row {b}: ring output differs"
            assert logits_close(tl[b], g[f"{name}_text_logits"][s, b], wd), f"{name} step {s} row {b}: text logits"
            for k in range(cfg.dep_q):
                assert logits_close(al[b, k], g[f"{name}_audio_logits"][s, b, k], wd), f"{name} step {s} row {b} cb {k}: audio logits"


def check_cross_engine(device, lib, name):
    """The engine's cross-attention path against the reference's run: LMModel(cross_attention) + ConditionFuser(cross) + LMGen."""
    g, cfg, sd = load_cross_golden()
    sc = CROSS_SCENARIOS[name]
    B = g["masks"].shape[1]
    rows = 2 * B if sc["cfg_coef"] != 1.0 else B
    conds, fuser = cross_condition_tensors(g, name)
    lm = LMModel(sd, cfg, device=device, max_batch=rows, lib=lib, fuser=fuser)
    state = {}

    def start(cfg_coef, condition_sum, condition_cross):
        gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, cfg_coef=cfg_coef, condition_tensors=conds)
        gen.streaming_forever(B)
        state["gen"] = gen

    def step(codes, forced, mask, reset):
        gen = state["gen"]
        if reset is not None:
            gen.reset_streaming(torch.from_numpy(reset).to(device))
        gen.set_exec_mask(torch.from_numpy(mask).to(device))
        out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
        return out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
    try:
        check_cross_scenario(g, cfg, name, start, step)
    finally:
        if "gen" in state:
            state["gen"]._stop_streaming()


def cross_vs_oracle(device, lib, cfg, B, S, Tc, seed):
    """Engine against the oracle on a cross-attention model of any width (the golden covers the tiny one): T_c condition
    positions, rows at different depths of the ring."""
    from moshi_amd.lm import ConditionFuser
    sd = random_lm_state_dict(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    cx = (0.7 * torch.randn(B, Tc, cfg.dim, generator=g)).to(torch.bfloat16)
    conds = {"x": (cx, torch.ones(B, Tc, dtype=torch.bool))}
    lm = LMModel(sd, cfg, device=device, max_batch=B, lib=lib, fuser=ConditionFuser({"sum": [], "cross": ["x"]}))
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, condition_tensors=conds)
    orc = LMOracle(sd, cfg)
    orc.streaming(B, condition_cross=cx.float().numpy())
    rng = np.random.default_rng(seed)
    with gen.streaming(B):
        for s in range(S):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            assert np.array_equal(out, oo), f"step {s}: ring output differs"
            for b in range(B):
                assert logits_close(tl[b], otl[b]), f"step {s} row {b}: text logits {np.abs(tl[b]-otl[b]).max()}"
                for k in range(cfg.dep_q):
                    assert logits_close(al[b, k], oal[b, k]), f"step {s} row {b} cb {k}"


# ---- fp8 on hardware ------------------------------------------------------------------------------------------------------------
# The gfx950 fp8 dot-product unit sums each group of 8 products with the small ones aligned to the largest (products below
# ~2^-13 of it are shifted out): scripts/fp8_probe.hip measures up to 2.7e-4 of sum|products| (the conversion v_cvt_pk_fp8_f32 is
# bit-exact).  A perturbation of that size flips bf16 roundings of a few % of the GEMM outputs, and every flip is re-quantised to
# e4m3 (6-12 % steps) by the next linear: the fp8 NETWORK is ill-conditioned at that level - the exact-accumulation oracle moved
# by 2.5e-4 deviates from ITSELF by max 9-15 % (median) / 19-28 % (worst), mean 3 % / 4-7 % of max|logit| (tiny / 7B-width).  So
# on hardware the fp8 engine is held to that yardstick, measured in the same test, instead of the bf16-level tolerance the
# exact-accumulation simulator meets.
FP8_HW_ACC_NOISE = 2.5e-4


def fp8_engine_within_format_conditioning(device, lib, cfg, seed, B, S, input_scale=1.0):
    from moshi_amd.weights import quantize_lm_state_dict_fp8
    bf = cached_lm_state_dict(cfg, seed)
    sd = quantize_lm_state_dict_fp8(bf, input_scale=input_scale)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    exact, moved, plain = LMOracle(sd, cfg), LMOracle(sd, cfg, fp8_accumulate_noise=FP8_HW_ACC_NOISE, noise_seed=seed), LMOracle(bf, cfg)
    for o in (exact, moved, plain):
        o.streaming(B)
    rng = np.random.default_rng(seed)
    dev = {"engine": [], "self": [], "engine_vs_bf16": [], "oracle_vs_bf16": []}

    def add(key, a, ref):
        sc = float(np.abs(ref).max()) + 1e-6
        d = np.abs(a - ref)
        dev[key].append((float(d.max()) / sc, float(d.mean()) / sc))
    with gen.streaming(B):
        for s in range(S):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = exact.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            _, (mtl, mal, _, _) = moved.step(codes, use_sampling=False, forced=forced, support_out_of_sync=True)
            _, (ptl, pal, _, _) = plain.step(codes, use_sampling=False, forced=forced, support_out_of_sync=True)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            assert np.array_equal(out, oo), f"step {s}: ring output differs"        # integer bookkeeping stays exact
            for b in range(B):
                sites = [(tl[b], otl[b], mtl[b], ptl[b])] + [(al[b, k], oal[b, k], mal[b, k], pal[b, k]) for k in range(cfg.dep_q)]
                for e, o, m, p_ in sites:
                    add("engine", e, o); add("self", m, o); add("engine_vs_bf16", e, p_); add("oracle_vs_bf16", o, p_)
    st = {k: (float(np.median([v[0] for v in vs])), float(max(v[0] for v in vs)), float(np.median([v[1] for v in vs])), float(max(v[1] for v in vs)))
          for k, vs in dev.items()}      # (median max-rel, worst max-rel, median mean-rel, worst mean-rel)
    msg = " | ".join(f"{k}: max-rel {v[0]:.3f}/{v[1]:.3f} mean-rel {v[2]:.3f}/{v[3]:.3f}" for k, v in st.items())
    e, f = st["engine"], st["self"]
    assert e[0] <= 1.25 * f[0] + 0.01 and e[1] <= 1.5 * f[1] + 0.02, "engine further from the oracle than the format's own conditioning: " + msg
    assert e[2] <= 1.25 * f[2] + 0.005 and e[3] <= 1.5 * f[3] + 0.01, "engine further from the oracle than the format's own conditioning: " + msg
    a, r = st["engine_vs_bf16"], st["oracle_vs_bf16"]
    assert a[2] <= 1.5 * r[2] + 0.005 and a[3] <= 1.75 * r[3] + 0.01, "fp8 engine less accurate against the bf16 model than the fp8 oracle: " + msg
    return msg


# ---- ASR-style models: dep_q = 0, no depformer (lm.py:218-221), text delayed behind the audio, extra heads --------------------------
def check_stt_golden(start, step, heads):
    """tests/golden/lm_stt.npz (reference run of moshi_amd.config.tiny_stt_config).  start() opens a stream of B sessions;
    step(codes [B, 8, 1], forced [B, 1], mask, reset_or_None) -> (tokens [B, 1, 1], text logits [B, V]); heads() -> [B, 2, 6]."""
    g = np.load(GOLDEN / "lm_stt.npz")
    S, B = g["masks"].shape
    start()
    for s in range(S):
        reset = g["reset_mask"] if s == int(g["reset_step"][0]) else None
        out, tl = step(g["codes"][s], g["text_tok"][s][:, None], g["masks"][s], reset)
        pr = heads()
        for b in range(B):
            if not g["masks"][s, b]:
                continue
            assert np.array_equal(out[b], g["tokens"][s, b]), f"step {s} row {b}: ring output differs"
            assert logits_close(tl[b], g["text_logits"][s, b]), f"step {s} row {b}: text logits"
            assert np.abs(pr[b] - g["heads"][s, b]).max() <= 0.02, f"step {s} row {b}: extra heads"


def check_stt_engine(device, lib):
    from moshi_amd.config import tiny_stt_config
    g = np.load(GOLDEN / "lm_stt.npz")
    cfg = tiny_stt_config()
    sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))
    B = g["masks"].shape[1]
    lm = LMModel(sd, cfg, device=device, max_batch=B, lib=lib)
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    assert lm.dep_q == 0

    def step(codes, forced, mask, reset):
        if reset is not None:
            gen.reset_streaming(torch.from_numpy(reset).to(device))
        gen.set_exec_mask(torch.from_numpy(mask).to(device))
        out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
        assert out.shape == (B, 1, 1) and al.shape == (B, 0, cfg.card)
        return out.cpu().numpy(), tl.cpu().numpy()

    def heads():
        probs = torch.empty(B, cfg.extra_heads_num_heads, cfg.extra_heads_dim, device=device, dtype=torch.float32)
        gen._lib.check(gen._lib.mmi_lm_extra_heads(lm._handle, probs.data_ptr(), gen._stream()))
        return probs.cpu().numpy()
    try:
        check_stt_golden(lambda: gen.streaming_forever(B), step, heads)
        # the public call: tokens + one probability tensor per head
        out, hs = gen.step_with_extra_heads(torch.from_numpy(g["codes"][0]).to(device))
        assert out.shape == (B, 1, 1) and len(hs) == 2 and hs[0].shape == (B, 1, cfg.extra_heads_dim)
    finally:
        gen._stop_streaming()


def check_streaming_state_snapshot(device, lib):
    """StreamingModule.get_streaming_state / set_streaming_state (streaming.py:158-181): a snapshot taken mid-dialogue and
    loaded back reproduces the continuation bit for bit (codec and LM; greedy)."""
    from tests import batcher_cases
    B = 2
    mimi, lm, mcfg, lcfg = batcher_cases.tiny_pair(device, lib, B)
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    rng = np.random.default_rng(5)
    frames = [torch.from_numpy((0.3 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32)).to(device) for _ in range(5)]

    def run(fs):
        out = []
        for x in fs:
            tokens = gen.step(mimi.encode(x))
            out.append((tokens.clone(), mimi.decode(tokens[:, 1:].clamp(min=0)).clone()))
        return out
    with mimi.streaming(B), gen.streaming(B):
        run(frames[:2])
        snap_m, snap_l = mimi.get_streaming_state(), gen.get_streaming_state()
        first = run(frames[2:])
        run(frames[:1])                                     # wander off
        mimi.set_streaming_state(snap_m); gen.set_streaming_state(snap_l)
        again = run(frames[2:])
        try:
            gen.set_streaming_state({})
            raise AssertionError("an empty state must be refused")
        except RuntimeError as e:
            assert "streaming state" in str(e)
    for (ta, pa), (tb, pb) in zip(first, again):
        assert torch.equal(ta, tb) and torch.equal(pa, pb)


# ---- top_k = 0: multinomial over the whole vocabulary, one Exp(1) per entry from the engine's counter RNG ----------------------------
def philox_exp_noise(seed: int, step: int, a: int, idx: np.ndarray) -> np.ndarray:
    """lm_kernels.h mmi_exp_noise: Philox4x32-10 with counter (step lo, step hi, a, idx), key = seed -> -log(u)."""
    M = np.uint64(0xFFFFFFFF)
    c0 = np.full(idx.shape, step & 0xFFFFFFFF, np.uint64); c1 = np.full(idx.shape, (step >> 32) & 0xFFFFFFFF, np.uint64)
    c2 = np.full(idx.shape, a, np.uint64); c3 = idx.astype(np.uint64)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    u = ((c0 >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return (-np.log(u)).astype(np.float32)


def check_full_multinomial(device, lib, steps=4, B=3, seed=77):
    """LMGen(top_k=0, top_k_text=0): every sampled token equals argmax(softmax(logits / temp) / q) over the WHOLE vocabulary with
    q the engine's own per-entry draws (sampling.py:40-47, 98-106), recomputed here from the logits taps."""
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=21)
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=True, temp=0.9, temp_text=0.8, top_k=0, top_k_text=0, seed=seed,
                      support_out_of_sync=True)
    rng = np.random.default_rng(1)

    def pick(logits, temp, step, site):
        out = np.zeros(B, np.int64)
        for b in range(B):
            x = (logits[b] / np.float32(temp)).astype(np.float32)
            pr = np.exp(x - x.max()).astype(np.float32)
            pr = pr / pr.sum(dtype=np.float32)
            out[b] = int(np.argmax(pr / philox_exp_noise(seed, step, site * B + b, np.arange(logits.shape[1]))))
        return out
    prev = None
    with gen.streaming(B):
        for s in range(steps):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device))
            out, tl, al = out.cpu().numpy()[:, :, 0], tl.cpu().numpy(), al.cpu().numpy()
            text = pick(tl, 0.8, s, 0)
            audio = np.stack([pick(al[:, k], 0.9, s, 1 + k) for k in range(cfg.dep_q)], 1)
            if prev is not None:      # the ring returns text and codebook 0 one step late (delays 0), the others at once (delays 1)
                assert np.array_equal(out[:, 0], prev[0]) and np.array_equal(out[:, 1], prev[1][:, 0]), f"step {s}"
                assert np.array_equal(out[:, 2:], audio[:, 1:]), f"step {s}"
            prev = (text, audio)


def philox4_words(seed: int, step: int, a: int, idx: np.ndarray) -> np.ndarray:
    """lm_kernels.h mmi_philox4: the four output words of Philox4x32-10 at counter (step lo, step hi, a, idx), key = seed: [n, 4] uint64."""
    M = np.uint64(0xFFFFFFFF)
    c0 = np.full(idx.shape, step & 0xFFFFFFFF, np.uint64); c1 = np.full(idx.shape, (step >> 32) & 0xFFFFFFFF, np.uint64)
    c2 = np.full(idx.shape, a, np.uint64); c3 = idx.astype(np.uint64)
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & M, p1 & M, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & M, p0 & M
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & M, (k1 + np.uint64(0xBB67AE85)) & M
    return np.stack([c0, c1, c2, c3], -1)


def check_topk_device_rng(device, lib, cfg=None, top_k=20, top_k_text=10, steps=4, B=3, seed=77):
    """The sampler's PRODUCTION form (top-k, the engine's own counter RNG, no supplied noise), token for token: the top-k set of the
    bf16 logits (ties at the threshold towards the lower index), each member scored logit / temp - log(q) with q = -log(u) and u
    from word (i & 3) of Philox4x32-10 at counter (step, site * B + session, i >> 2), the largest score wins (lm_kernels.h
    k_sample, fast path) - recomputed here from the logits taps.  A different token is accepted only where the two best scores are
    closer than the device's logf can tell apart."""
    cfg = cfg or tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=21)
    temp, temp_text = 0.9, 0.8
    gen = make_engine(cfg, sd, device, lib, B, use_sampling=True, temp=temp, temp_text=temp_text, top_k=top_k, top_k_text=top_k_text,
                      seed=seed, support_out_of_sync=True)
    rng = np.random.default_rng(1)
    near = [0]

    def pick(logits, t, k, step, site, got):
        for b in range(B):
            bits = (logits[b].astype(np.float32).view(np.uint32) >> 16).astype(np.uint32)
            key = np.where(bits & 0x8000, (~bits) & 0xFFFF, bits | 0x8000).astype(np.int64)
            V = len(key)
            order = np.lexsort((np.arange(V), -key))[:min(k, V)]
            words = philox4_words(seed, step, site * B + b, order >> 2)[np.arange(len(order)), order & 3]
            u = ((words >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
            sc = (logits[b][order].astype(np.float32) * (np.float32(1.0) / np.float32(t))).astype(np.float32) - np.log(-np.log(u)).astype(np.float32)
            best = order[np.lexsort((order, -sc))]
            if int(got[b]) != int(best[0]):
                s0 = float(sc[order == best[0]][0])
                assert int(got[b]) in order.tolist(), f"step {step} site {site} row {b}: token {int(got[b])} is not in the top-{k} set"
                s1 = float(sc[order == int(got[b])][0])
                assert abs(s0 - s1) <= 2e-5 * max(1.0, abs(s0)), f"step {step} site {site} row {b}: token {int(got[b])} (score {s1}) instead of {int(best[0])} ({s0})"
                near[0] += 1
    prev = None
    with gen.streaming(B):
        for s in range(steps):
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device))
            out, tl, al = out.cpu().numpy()[:, :, 0], tl.cpu().numpy(), al.cpu().numpy()
            if prev is not None:      # the ring returns text and codebook 0 one step late (delays 0), the others at once (delays 1)
                pick(prev[0], temp_text, top_k_text, s - 1, 0, out[:, 0])
                pick(prev[1][:, 0], temp, top_k, s - 1, 1, out[:, 1])
                for kq in range(1, cfg.dep_q):
                    pick(al[:, kq], temp, top_k, s, 1 + kq, out[:, 1 + kq])
            prev = (tl, al)
    assert near[0] <= 2, f"{near[0]} sampled tokens sat on a near-tie of the two best scores"


# ---- C5, per linear, bit for bit ("the int8 GEMM is integer work") ------------------------------------------------
# `QLinear.forward` (utils/quantize.py:24-40) of ONE module on the same bf16 rows, engine (`mmi_lm_debug_linear`: the kernels the
# step uses) against the oracle's restatement of bitsandbytes' rule: the int8 codes, the row absmax and the bf16 output must be
# IDENTICAL - row-wise quantisation, an exact int32 dot product and one fp32 expression `out32 * (SCA * SCB) * (1 / 127^2)` leave
# no room for rounding-order differences.  Gated linears: the SiLU's exp is a transcendental (device expf vs numpy exp), so their
# gated output is held to ">= 99.9 % of the elements identical, the others one bf16 ulp apart"; their integer part is the same
# kernel code as the un-gated linears'.
def _dyadic_rows(rng, B, K):
    """Rows whose sum of squares is exact in fp32 in ANY summation order (entries +-2^-2 .. +-2^1): the RMSNorm in front of a
    linear is then one well-defined function, and the norm-fused kernels can be held to bit equality too."""
    mag = 2.0 ** rng.integers(-2, 2, (B, K)).astype(np.float32)
    return (mag * rng.choice(np.array([-1.0, 1.0], np.float32), (B, K))).astype(np.float32)


def _bf16_bits(a: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(a, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)


def int8_linears_bit_exact(device, lib, cfg, B, seed, report=None, weights_seed=None):
    from moshi_amd.weights import quantize_lm_state_dict
    from oracle.lm_oracle import QWeight, bf16r, int8_vectorwise_quant, linear_int8, rms_norm, silu
    sd = quantize_lm_state_dict(cached_lm_state_dict(cfg, weights_seed if weights_seed is not None else seed))
    lm = LMModel(sd, cfg, device=device, max_batch=B, lib=lib)
    rng = np.random.default_rng(seed)
    d, dd = cfg.dim, cfg.depformer_dim
    k_last = cfg.dep_q - 1
    # (weight key, the norm vector in front of it in the model or None, paths)
    sites = [
        ("transformer.layers.0.self_attn.in_projs.0.weight", "transformer.layers.0.norm1.alpha", ["plain", "norm"]),
        ("transformer.layers.0.self_attn.out_projs.0.weight", None, ["plain", "splitk"]),
        ("transformer.layers.0.gating.linear_in.weight", "transformer.layers.0.norm2.alpha", ["plain", "norm"]),
        ("transformer.layers.0.gating.linear_out.weight", None, ["plain", "splitk"]),
        ("text_linear.weight", "out_norm.alpha", ["plain", "norm"]),
        ("depformer_in.1.weight", None, ["plain"]),
        (f"depformer.layers.0.self_attn.in_projs.{k_last}.weight", "depformer.layers.0.norm1.alpha", ["plain", "norm", "norm_fused"]),
        ("depformer.layers.1.self_attn.out_projs.0.weight", None, ["plain", "fused"]),
        (f"depformer.layers.1.gating.{k_last}.linear_in.weight", "depformer.layers.1.norm2.alpha", ["plain", "norm", "norm_fused"]),
        ("depformer.layers.0.gating.2.linear_out.weight", None, ["plain", "fused"]),
        ("linears.3.weight", None, ["plain", "fused"]),
    ]
    checked = []
    for key, alpha_key, paths in sites:
        w = QWeight(sd[key].numpy(), sd[key + "_scb"].numpy())
        w.act8 = True
        K = w.q.shape[1]
        gated = "linear_in" in key
        alpha = None if alpha_key is None else bf16r(sd[alpha_key].float().numpy().reshape(-1))

        def oracle_out(rows):
            h = linear_int8(rows, w)
            if not gated:
                return h
            H = h.shape[1] // 2
            return bf16r(bf16r(silu(h[:, :H])) * h[:, H:])

        for path in paths:
            if path in ("norm", "norm_fused"):
                x = _dyadic_rows(rng, B, K)
            else:           # heavy-tailed rows, a zero row and a row whose absmax sits on one entry
                x = bf16r((rng.standard_normal((B, K)) * np.exp(rng.standard_normal((B, 1)))).astype(np.float32))
                x[B - 1] = 0.0
                if B > 1:
                    x[0, rng.integers(0, K)] = 37.0
            xt = torch.from_numpy(x)
            try:
                res = lm.debug_linear(key, xt, path=path, alpha_name=alpha_key if path in ("norm", "norm_fused") else None,
                                      want_codes=path in ("plain", "splitk", "norm"))
            except NotImplementedError as e:
                if path == "splitk" and "does not split" in str(e):     # tiny shapes are not split over K unless forced
                    continue
                raise
            out = res["out"].float().cpu().numpy()
            name = f"{key} [{path}] B={B}"
            rows = x
            if path == "norm":        # the engine's norm output must be the oracle's, and it is what the linear quantises
                yn = res["norm"].float().cpu().numpy()
                assert np.array_equal(_bf16_bits(yn), _bf16_bits(rms_norm(x, alpha))), f"{name}: RMSNorm output differs on exact-sum rows"
                rows = yn
            elif path == "norm_fused":
                rows = rms_norm(x, alpha)
            if "codes" in res:
                ca, sca = int8_vectorwise_quant(rows)
                assert np.array_equal(res["absmax"].cpu().numpy(), sca[:, 0]), f"{name}: row absmax differs"
                assert np.array_equal(res["codes"].cpu().numpy().astype(np.int32), ca.astype(np.int32)), f"{name}: int8 codes differ"
            ref = oracle_out(rows)
            same = _bf16_bits(out) == _bf16_bits(ref)
            if gated:
                ulp = np.abs(out - ref) <= np.maximum(np.abs(ref), 1e-30) * 2.0 ** -7
                assert same.mean() >= 0.999 and ulp.all(), f"{name}: gated output {same.mean():.5f} identical, worst {np.abs(out - ref).max()}"
            else:
                assert same.all(), f"{name}: {int((~same).sum())} of {same.size} bf16 outputs differ (first at {np.argwhere(~same)[0]})"
            checked.append((name, float(same.mean())))
    if report is not None:
        report.extend(checked)
    return checked


# ---- C5 at network level: a statistical gate with a yardstick ------------------------------------------------------
# With every int8 linear bit-identical to the oracle for the same input rows (int8_linears_bit_exact), engine and oracle can
# only part ways at the pieces that are NOT integer work - RMSNorm's mean of squares and rsqrt, the softmax of the attentions,
# SiLU, RoPE - where summation order / the device's exp move a bf16 result by one rounding flip now and then.  Re-quantising
# every activation row to 8 bits is a discontinuous map: most flips vanish (the int8 codes do not change: the (row, site) pair
# is bit-identical to the oracle), a few are carried to the logits, and a flip of a row's absmax entry moves the whole row.  The
# error per (row, site) is therefore heavy-tailed, and a threshold on its MAXIMUM over a few thousand pairs is a coin that
# eventually lands wrong - it did, at the driver, in round 4.  The gate: (1) token-ring outputs exact under teacher forcing;
# (2) the share of pairs inside the plain bf16 tolerance, the median and the 90th percentile of the error are no worse than
# those of ANOTHER correct implementation of the same network (the oracle with its norm / softmax / SiLU statistics in fp64,
# `LMOracle(stat64=True)`), with a margin; (3) no pair is grossly wrong (a broken kernel gives errors of order max|logit|).
INT8_NET_GROSS_MAX, INT8_NET_GROSS_MEAN = 0.35, 0.10


def int8_network_vs_oracle(device, lib, cfg, seed, B, S, use_masks=True, name=None, stats=None, on_device_draw=False):
    from moshi_amd.weights import quantize_lm_state_dict
    if on_device_draw:
        # the benchmark model (32 layers, 7.7 B parameters): drawn and quantised on the GPU (seconds instead of minutes on the
        # host), and the two checkers SHARE one host copy of the int8 weights (the yardstick differs in its statistics only)
        import copy
        sd = quantize_lm_state_dict(random_lm_state_dict(cfg, seed=seed, device=device))
        gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
        orc = LMOracle(sd, cfg)
        yard = copy.copy(orc)
        yard.stat64 = True
    else:
        sd = quantize_lm_state_dict(cached_lm_state_dict(cfg, seed))
        gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
        orc, yard = LMOracle(sd, cfg), LMOracle(sd, cfg, stat64=True)
    del sd
    orc.streaming(B); yard.streaming(B)
    rng = np.random.default_rng(seed)
    eng, yd = [], []

    def rel(a, ref):
        sc = float(np.abs(ref).max()) + 1e-6
        dlt = np.abs(a - ref)
        return float(dlt.max()) / sc, float(dlt.mean()) / sc
    with gen.streaming(B):
        for s in range(S):
            mask = np.ones(B, bool)
            if use_masks and B > 1:
                mask = rng.random(B) > 0.3
                mask[0] = True
                if s == S // 2:
                    r = np.zeros(B, bool); r[B - 1] = True
                    for o in (orc, yard):
                        o.reset_streaming(r)
                    gen.reset_streaming(torch.from_numpy(r).to(device))
                    mask[B - 1] = True
            for o in (orc, yard):
                o.set_exec_mask(mask)
            gen.set_exec_mask(torch.from_numpy(mask).to(device))
            codes = rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            _, (ytl, yal, _, _) = yard.step(codes, use_sampling=False, support_out_of_sync=True, forced=forced)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            for b in range(B):
                if not mask[b]:
                    continue
                assert np.array_equal(out[b], oo[b]), f"step {s} row {b}: ring output differs"
                eng.append(rel(tl[b], otl[b])); yd.append(rel(ytl[b], otl[b]))
                for k in range(cfg.dep_q):
                    eng.append(rel(al[b, k], oal[b, k])); yd.append(rel(yal[b, k], oal[b, k]))
        if stats is not None:
            import collections
            stats["xlds_launches"] = int(gen._lib.mmi_lm_stat(gen.lm_model._handle, 0))
            stats["launch_sites"] = dict(collections.Counter(site for site, _ in gen.launch_list()))
            stats["launch_list"] = list(gen.launch_list())
            stats["dep_tile"] = int(gen._lib.mmi_lm_stat(gen.lm_model._handle, 3))
    e, y = np.array(eng), np.array(yd)

    def summ(v):
        inside = (v[:, 0] <= LOGIT_MAX_REL) & (v[:, 1] <= LOGIT_MEAN_REL)
        return {"pairs": int(len(v)), "identical": float((v[:, 0] == 0).mean()), "inside_bf16_tolerance": float(inside.mean()),
                "max_rel_median": float(np.median(v[:, 0])), "max_rel_p90": float(np.quantile(v[:, 0], 0.9)),
                "max_rel_p99": float(np.quantile(v[:, 0], 0.99)), "max_rel_worst": float(v[:, 0].max()),
                "mean_rel_median": float(np.median(v[:, 1])), "mean_rel_p90": float(np.quantile(v[:, 1], 0.9)), "mean_rel_worst": float(v[:, 1].max())}
    es, ys = summ(e), summ(y)
    res = {"case": name or f"int8_network_b{B}", "engine_vs_oracle": es, "yardstick_fp64_statistics_vs_oracle": ys}
    print(f"[parity] {res['case']}: engine {es}\n[parity] {res['case']}: yardstick {ys}")
    out_dir = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out_dir.is_dir() and name:
        import json
        (out_dir / f"parity_{name}.json").write_text(json.dumps(res, indent=1))
    assert es["inside_bf16_tolerance"] >= min(0.85, ys["inside_bf16_tolerance"] - 0.10), f"too few (row, site) pairs inside the bf16 tolerance: {es} vs yardstick {ys}"
    # quantile against quantile, the yardstick's taken 5 points higher: at full depth ~90 % of the pairs are BIT-IDENTICAL to the oracle
    # on both sides (engine 87.5 %, yardstick 91.6 % at 32 layers / 64 sessions), so a percentile can be exactly 0 for one and the
    # first non-identical pair for the other - the comparison must tolerate a few points of difference in where the identical block ends
    for col, what, q, slack in ((0, "max_rel", 0.5, 0.005), (0, "max_rel", 0.9, 0.01), (0, "max_rel", 0.99, 0.02), (1, "mean_rel", 0.5, 0.002), (1, "mean_rel", 0.9, 0.004)):
        eq, yq = float(np.quantile(e[:, col], q)), float(np.quantile(y[:, col], min(q + 0.05, 1.0)))
        assert eq <= 1.5 * yq + slack, f"{what} quantile {q}: engine {eq:.4f} vs yardstick's quantile {min(q + 0.05, 1.0):.2f} {yq:.4f}"
    assert es["identical"] >= ys["identical"] - 0.10, f"share of pairs bit-identical to the oracle: engine {es['identical']:.3f} vs yardstick {ys['identical']:.3f}"
    assert es["max_rel_worst"] <= INT8_NET_GROSS_MAX and es["mean_rel_worst"] <= INT8_NET_GROSS_MEAN, f"a (row, site) pair is grossly wrong: {es}"
    return res


# ---- C5 against the PINNED reference ("C5 parity = against the bf16 oracle with a stated quantisation tolerance") ------
# The int8 rule itself cannot be pinned (bitsandbytes is not in /root/reference and cannot run here), but the distance of the int8
# ENGINE to the reference's own bf16 logits can be measured: the goldens lm_wide.npz (7B layer shapes, 1 temporal layer) and
# lm_full.npz (the benchmark model, 32 layers) are outputs of the reference itself.  The int8 network is a different function than
# the bf16 one (weights and activations rounded to 8 bits row by row), so the gate has two parts: a STATED quantisation tolerance
# per (row, site) pair - max |d| <= 25 % and mean |d| <= 6 % of max|logit| of the reference, what row-wise int8 costs a random-init
# network at these widths, read off the int8 ORACLE's own distance to the same goldens - and, the part that binds, the engine no
# further from the reference than 1.5 x the int8 oracle is, statistic by statistic.
INT8_VS_BF16_MAX_REL = 0.25
INT8_VS_BF16_MEAN_REL = 0.06


def int8_engine_vs_bf16_reference(device, lib, which="wide", max_batch=None, name=None, sd=None):
    from dataclasses import replace
    from moshi_amd.weights import quantize_lm_state_dict
    g, cfg = load_wide() if which == "wide" else load_full()
    if sd is None:
        sd = random_lm_state_dict(cfg, seed=int(g["seed"][0]))          # the generator's (CPU) draw
    sdq = quantize_lm_state_dict(sd)
    del sd
    S, B = g["g_text_tok"].shape
    gen = make_engine(cfg, sdq, device, lib, max_batch or B, use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sdq, replace(cfg, context=min(cfg.context, 64)))      # (no wrap inside the golden's few steps)
    del sdq
    orc.streaming(B)
    eng, ora = [], []

    def rel(a, ref):
        sc = float(np.abs(ref).max()) + 1e-6
        dlt = np.abs(a - ref)
        return float(dlt.max()) / sc, float(dlt.mean()) / sc
    with gen.streaming(B):
        for s in range(S):
            mask = g["masks"][s] if "masks" in g else np.ones(B, bool)
            gen.set_exec_mask(torch.from_numpy(mask).to(device))
            orc.set_exec_mask(mask)
            forced = np.concatenate([g["g_text_tok"][s][:, None], g["g_audio_tok"][s]], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(g["codes"][s]).to(device), forced_tokens=torch.from_numpy(forced).to(device))
            out, tl, al = out.cpu().numpy(), tl.cpu().numpy(), al.cpu().numpy()
            oo, (otl, oal, _, _) = orc.step(g["codes"][s], use_sampling=False, support_out_of_sync=True, forced=forced)
            for b in range(B):
                if not mask[b]:
                    assert (out[b] == -2).all()
                    continue
                assert np.array_equal(out[b], g["g_tokens"][s, b]), f"step {s} row {b}: ring output differs from the reference's"
                assert np.array_equal(out[b], oo[b])
                eng.append(rel(tl[b], g["g_text_logits"][s, b])); ora.append(rel(otl[b], g["g_text_logits"][s, b]))
                for k in range(cfg.dep_q):
                    eng.append(rel(al[b, k], g["g_audio_logits"][s, b, k])); ora.append(rel(oal[b, k], g["g_audio_logits"][s, b, k]))
    e, o = np.array(eng), np.array(ora)

    def summ(v):
        inside = (v[:, 0] <= INT8_VS_BF16_MAX_REL) & (v[:, 1] <= INT8_VS_BF16_MEAN_REL)
        return {"pairs": int(len(v)), "inside_quantisation_tolerance": float(inside.mean()),
                "max_rel_median": float(np.median(v[:, 0])), "max_rel_p90": float(np.quantile(v[:, 0], 0.9)), "max_rel_worst": float(v[:, 0].max()),
                "mean_rel_median": float(np.median(v[:, 1])), "mean_rel_p90": float(np.quantile(v[:, 1], 0.9)), "mean_rel_worst": float(v[:, 1].max())}
    es, os_ = summ(e), summ(o)
    case = name or f"c5_int8_vs_reference_{which}"
    res = {"case": case, "golden": f"lm_{which}.npz (the reference's bf16 logits)", "tolerance": [INT8_VS_BF16_MAX_REL, INT8_VS_BF16_MEAN_REL],
           "int8_engine_vs_bf16_reference": es, "int8_oracle_vs_bf16_reference": os_}
    print(f"[parity] {case}: engine {es}\n[parity] {case}: int8 oracle {os_}")
    out_dir = Path(__file__).resolve().parent.parent / "gpurun_out"
    if out_dir.is_dir():
        import json
        (out_dir / f"parity_{case}.json").write_text(json.dumps(res, indent=1))
    assert es["inside_quantisation_tolerance"] >= min(0.9, os_["inside_quantisation_tolerance"] - 0.05), f"{case}: too few pairs inside the quantisation tolerance: {es} vs the int8 oracle {os_}"
    for key, slack in (("max_rel_median", 0.005), ("max_rel_p90", 0.01), ("mean_rel_median", 0.002), ("mean_rel_p90", 0.004)):
        assert es[key] <= 1.5 * os_[key] + slack, f"{case} {key}: engine {es[key]:.4f} vs the int8 oracle's {os_[key]:.4f} from the same reference"
    return res


# ---- the step is a function of its inputs: two streams of one handle fed the same frames produce the same bits ----------------------
def reproducible_between_streams(device, lib, cfg, B, quantize=False, steps=3, repeats=4, seed=77, before_repeat=None, guided_cross=None):
    """Round 5's root cause (the driver's round-4 failure) was a launch whose output changed from run to run; what caught it was
    comparing runs, not comparing with the checker.  Greedy, masks and a partial reset, hidden taps on: every repeat must equal the
    first in tokens, text / audio logits and the residual stream after the first and the last temporal layer, bit for bit."""
    sd = cached_lm_state_dict(cfg, seed)
    if quantize == "fp8":
        from moshi_amd.weights import quantize_lm_state_dict_fp8
        sd = quantize_lm_state_dict_fp8(sd)
    elif quantize:
        from moshi_amd.weights import quantize_lm_state_dict
        sd = quantize_lm_state_dict(sd)
    if guided_cross:            # (T_c, coef): a cross-attention model under classifier-free guidance - two model rows per session, the
        from moshi_amd.lm import ConditionFuser     # norm_cross / query GEMM / cross-attention / out_proj launches, k_cfg_mix at every site
        Tc, coef = guided_cross
        g = torch.Generator().manual_seed(seed)
        cx = (0.7 * torch.randn(2 * B, Tc, cfg.dim, generator=g)).to(torch.bfloat16)
        lm = LMModel(sd, cfg, device=device, max_batch=2 * B, lib=lib, fuser=ConditionFuser({"sum": [], "cross": ["x"]}))
        gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, cfg_coef=coef,
                    condition_tensors={"x": (cx, torch.ones(2 * B, Tc, dtype=torch.bool))})      # conditioned rows first, then the null twins
    else:
        gen = make_engine(cfg, sd, device, lib, B, use_sampling=False, support_out_of_sync=True)
    gen.lm_model.enable_hidden_taps()
    rng = np.random.default_rng(seed)
    plan = []
    for s in range(steps):
        mask = rng.random(B) > 0.3
        mask[0] = True
        reset = None
        if s == 1 and B > 1:
            reset = np.zeros(B, bool); reset[B - 1] = True
        plan.append((mask, reset, rng.integers(0, cfg.card, (B, cfg.n_q - cfg.dep_q, 1))))

    def run():
        out = []
        with gen.streaming(B):
            for mask, reset, codes in plan:
                if reset is not None:
                    gen.reset_streaming(torch.from_numpy(reset).to(device))
                gen.set_exec_mask(torch.from_numpy(mask).to(device))
                o, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(device))
                out.append((o.cpu(), tl.cpu(), al.cpu(), gen.hidden_taps().cpu()))
        return out
    first = run()
    for r in range(repeats):
        if before_repeat is not None:       # the simulator's schedule test changes the fiber schedule between repeats
            before_repeat(r)
        again = run()
        for s, (a, b) in enumerate(zip(first, again)):
            m = torch.from_numpy(plan[s][0])
            for name, x, y in zip(("tokens", "text logits", "audio logits"), a[:3], b[:3]):
                assert torch.equal(x[m], y[m]), f"repeat {r} step {s}: {name} differ between two streams fed the same frames"
            mt = torch.cat([m, m]) if guided_cross else m          # guidance: the twins' rows follow the sessions'
            for w in (0, 1):
                rows = torch.nonzero((a[3][w][mt] != b[3][w][mt]).any(1)).flatten().tolist()
                assert not rows, f"repeat {r} step {s}: the residual stream after temporal layer {'0' if w == 0 else 'last'} differs in executing rows {rows[:8]}"
