"""Parity of the gfx950 LM engine on a real MI355X: reference golden vectors, the numpy oracle at tiny and at
full layer width, and size-independent properties at the benchmark batch."""
import numpy as np
import pytest
import torch

from moshi_amd.config import LMConfig, tiny_lm_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.weights import random_lm_state_dict
from tests import lm_cases

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_greedy_schedule_matches_reference_golden(gpu_lib):
    lm_cases.check_golden_greedy(DEV, None)


def test_7b_layer_shapes_match_reference_golden(gpu_lib):
    """Engine vs the reference's own output at Moshi-7B's real layer widths (tests/golden/lm_wide.npz)."""
    lm_cases.check_golden_wide(DEV, None)


def test_sampled_run_matches_reference_golden_given_its_noise(gpu_lib):
    lm_cases.check_golden_sampled(DEV, None)


def test_in_kernel_sampler_follows_the_reference_rule(gpu_lib):
    lm_cases.engine_sampling_matches_oracle_rule(DEV, None)


def test_full_size_sampler_follows_the_reference_rule(gpu_lib):
    """Moshi-7B vocabularies (32000 text / 2048 audio) and LMGen's default top-k 25 / 250 on the GPU."""
    cfg = LMConfig(num_layers=1, context=16)
    lm_cases.engine_sampling_matches_oracle_rule(DEV, None, cfg, top_k=250, top_k_text=25, B=5, steps=3)


@pytest.mark.parametrize("B", [1, 5, 18, 40])
def test_tiny_matches_oracle_with_masks_and_reset(gpu_lib, B):
    # B <= 16: 16x16x32 MFMA tile; 17..32: 32x32x16; 33..64: two batch tiles per weight fragment.  S > context=12: the ring wraps
    lm_cases.oracle_vs_engine(DEV, None, tiny_lm_config(), seed=50 + B, B=B, S=16 if B <= 5 else 4)


@pytest.mark.parametrize("kv,path", [("bf16", "launch"), ("bf16", "solo"), ("bf16", "switch"), ("bf16", "kernel"), ("fp8", "switch")])
def test_long_ring_split_over_workgroups_matches_oracle(gpu_lib, monkeypatch, kv, path):
    """A ring longer than one 256-slot chunk with few (session, head) pairs: the decode attention (k_lm_attn_wave) splits the
    ring over several workgroups and k_lm_attn_combine merges their partials ("launch": MMI_ATTN_SOLO=0 takes that step program
    from the first row) - or, while the ring is short ("solo": the default, up to 768 rows), workgroup 0 walks it alone and there
    is no merge launch; "switch": the engine goes from the "solo" program (graph) to the "launch" one after 100 steps, as a real
    session does after 768; "kernel": the shallow program's safety net, the merge done by the last workgroup to arrive inside the launch
    (agent-scope release / acquire across XCDs).  300 positions deep."""
    from dataclasses import replace
    from oracle.lm_oracle import LMOracle
    if path != "solo":
        monkeypatch.setenv("MMI_ATTN_SOLO", "100" if path == "switch" else "0")
    if path == "kernel":
        monkeypatch.setenv("MMI_ATTN_MERGE", "kernel")
    cfg = replace(tiny_lm_config(), context=600, kv_cache_dtype=kv)
    sd = random_lm_state_dict(cfg, seed=44)
    gen = LMGen(LMModel(sd, cfg, device=DEV, max_batch=1), use_sampling=False, support_out_of_sync=True)
    orc = LMOracle(sd, cfg)
    orc.streaming(1)
    rng = np.random.default_rng(44)
    with gen.streaming(1):
        for s in range(300):
            codes = rng.integers(0, cfg.card, (1, 8, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            if s % 20 and s < 290:       # the deep positions are what matters: skip the read-back on most steps
                gen._step(torch.from_numpy(codes).to(DEV), False, None, torch.from_numpy(forced).to(DEV))
                continue
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(DEV), forced_tokens=torch.from_numpy(forced).to(DEV))
            assert np.array_equal(out.cpu().numpy(), oo), f"step {s}"
            assert lm_cases.logits_close(tl[0].cpu().numpy(), otl[0]), f"step {s}: text logits"
            for k in range(cfg.dep_q):
                assert lm_cases.logits_close(al[0, k].cpu().numpy(), oal[0, k]), f"step {s} cb {k}"


@pytest.mark.parametrize("name", ["a", "b", "c", "d"])
def test_guidance_conditioning_and_extra_heads_match_reference_golden(gpu_lib, name):
    """SURVEY.md 8f-3 / 8f-4 on the GPU: classifier-free guidance (masked-until / no-text / condition tensors), a `sum`
    condition and extra heads against the reference's own LMGen runs (tests/golden/lm_cfg.npz)."""
    lm_cases.check_cfg_engine(DEV, None, name)


def test_sampling_without_top_k_is_a_multinomial_over_the_whole_vocabulary(gpu_lib):
    lm_cases.check_full_multinomial(DEV, None, steps=6, B=5)


def test_get_and_set_streaming_state_resume_a_dialogue(gpu_lib):
    lm_cases.check_streaming_state_snapshot(DEV, None)


def test_asr_style_model_without_depformer_matches_reference_golden(gpu_lib):
    lm_cases.check_stt_engine(DEV, None)


def test_guided_full_width_matches_oracle(gpu_lib):
    """Guidance at the 7B layer shapes (2 temporal layers): 2 x 3 model rows, masked-until + condition, vs the oracle."""
    from moshi_amd.lm import ConditionFuser
    from oracle.lm_oracle import LMOracle
    cfg = LMConfig(num_layers=2, context=64)
    sd = random_lm_state_dict(cfg, seed=12)
    B, S = 3, 3
    rng = np.random.default_rng(4)
    cond = torch.from_numpy(0.5 * rng.standard_normal((2 * B, 1, cfg.dim)).astype(np.float32)).to(torch.bfloat16)
    lm = LMModel(sd, cfg, device=DEV, max_batch=2 * B, fuser=ConditionFuser({"sum": ["c"]}))
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True, cfg_coef=2.0, cfg_is_masked_until=[0, 1, 2],
                condition_tensors={"c": (cond, None)})
    orc = LMOracle(sd, cfg)
    orc.streaming(B, cfg_coef=2.0, cfg_is_masked_until=[0, 1, 2], condition_sum=cond[:, 0].float().numpy())
    with gen.streaming(B):
        for s in range(S):
            codes = rng.integers(0, cfg.card, (B, 8, 1))
            oo, (otl, oal, ott, oat) = orc.step(codes, use_sampling=False, support_out_of_sync=True)
            forced = np.concatenate([ott[:, None], oat], 1)
            out, tl, al = gen.step_with_taps(torch.from_numpy(codes).to(DEV), forced_tokens=torch.from_numpy(forced).to(DEV))
            assert np.array_equal(out.cpu().numpy(), oo)
            for b in range(B):
                assert lm_cases.logits_close(tl[b].cpu().numpy(), otl[b], lm_cases.GUIDED_WIDEN), f"step {s} row {b}: text logits"
                for k in range(cfg.dep_q):
                    assert lm_cases.logits_close(al[b, k].cpu().numpy(), oal[b, k], lm_cases.GUIDED_WIDEN), f"step {s} row {b} cb {k}"


def test_full_width_layers_match_oracle(gpu_lib):
    """Moshi-7B layer shapes (dim 4096, 32 heads x 128, FFN 11264, text head 32000; depformer 1024 x 6 layers x 8 steps)
    with 2 temporal layers, so that the numpy oracle finishes in seconds: exercises every GEMM tile variant,
    the Dh=128 attention and the full-size sampler shapes used by the 7B model."""
    cfg = LMConfig(num_layers=2, context=64)
    lm_cases.oracle_vs_engine(DEV, None, cfg, seed=7, B=3, S=3, use_masks=True)


def test_full_depth_32_layers_at_the_benchmark_batch_match_oracle(gpu_lib):
    """The exact model `bench.py` times - 32 temporal layers, 3000-slot ring, 32 sessions - against the numpy oracle for three
    teacher-forced steps; the measured error per sampling site is printed and written to gpurun_out/parity_*.json."""
    lm_cases.full_depth_vs_oracle(DEV, None, B=32, S=2)      # (two steps: each widens 2 x 14.75 GB of weights to fp32 on the host)


def test_benchmark_model_matches_the_reference_at_full_depth(gpu_lib):
    """The model `bench.py` times (32 temporal layers, context 3000) against the REFERENCE's own LMGen run on the CPU
    (tests/golden/lm_full.npz, make_golden_lm_full.py): ring outputs identical, logits within FULL_WIDEN x the tolerance;
    measured distances in gpurun_out/parity_golden_full_cuda.json."""
    lm_cases.check_golden_full(DEV, None)


def test_benchmark_model_free_running_follows_the_reference(gpu_lib):
    """Greedy and NOT teacher-forced against the reference's own 32-layer run (lm_full.npz), on the benchmark's kernels (32-row
    tile): the step of first divergence per row is reported (gpurun_out/parity_golden_full_free_running.json) and must be a
    near-tie of the reference's logits."""
    lm_cases.check_golden_full_free_running(DEV, None, max_batch=32)


def test_benchmark_kernels_match_the_reference_at_full_depth(gpu_lib):
    """The same golden run on a handle built for 32 sessions: the 32-row tile, k_gemm_xlds, the split-K temporal GEMMs - the
    kernels `bench.py` times - against the reference's own logits (a 2-session handle takes the 16-row tile)."""
    lm_cases.check_golden_full(DEV, None, max_batch=32, name="golden_full_cuda_tile32")


def test_c5_int8_engine_against_the_reference_at_full_depth(gpu_lib):
    """C5 anchored to the PINNED reference at the benchmark's depth: the int8 x int8 engine (the golden's weights quantised row-wise)
    against the reference's own bf16 logits of its 32-layer run (tests/golden/lm_full.npz), within the stated quantisation tolerance
    and no further from them than 1.5 x the int8 oracle is - on a handle built for 64 sessions (two batch tiles, the C5 kernels).
    Here rather than in test_y_c5_int8_gpu.py because the benchmark model's 7.7 B parameters are drawn once for this file."""
    lm_cases.int8_engine_vs_bf16_reference(DEV, None, "full", max_batch=64, sd=lm_cases.full_golden_state_dict(DEV))
    lm_cases.release_full_golden_state_dict()       # the four tests above shared one host draw of the 7.7 B parameters


@pytest.mark.parametrize("B", [40, 64])
def test_two_batch_tiles_at_full_width_match_oracle(gpu_lib, B):
    """33..64 sessions at the 7B layer shapes on the DEFAULT GEMM path (k_gemm_xp with two batch tiles per weight fragment:
    what BASELINE configs[4] runs at 64 sessions)."""
    cfg = LMConfig(num_layers=2, context=64)
    lm_cases.oracle_vs_engine(DEV, None, cfg, seed=300 + B, B=B, S=2, use_masks=True)


@pytest.mark.parametrize("B", [18, 40])
def test_lds_resident_gemm_is_the_default_at_full_width(gpu_lib, B):
    """The default path of the wide temporal GEMMs at 17..64 sessions: k_gemm_xlds with the staggered tail (activations staged in
    LDS, one workgroup per CU walking 1-3 n-tiles, each tile's epilogue under the last chunk's weight stream), one and two
    batch tiles, against the oracle - and the engine reports that it took that kernel."""
    st = {}
    lm_cases.oracle_vs_engine(DEV, None, LMConfig(num_layers=2, context=64), seed=15, B=B, S=2, use_masks=False, stats=st)
    assert st["xlds_launches"] >= 2 * 2 + 1


def test_step_hooks_see_and_modify_the_step_like_the_reference(gpu_lib):
    lm_cases.check_step_hooks(DEV, None)


@pytest.mark.parametrize("B", [2, 4])
def test_ring_wraps_at_the_real_capacity(gpu_lib, B):
    """2 sessions: the ring split over 12 workgroups + merge launch; 4: one workgroup per (session, head) over all 3000 rows."""
    lm_cases.ring_wrap_at_real_capacity(DEV, None, B=B, S=14 if B == 2 else 7, seed=91 if B == 2 else 92)


def _greedy_run(cfg, sd, B, codes, steps, graph=True, monkeypatch=None):
    lm = LMModel(sd, cfg, device=DEV, max_batch=B)
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    outs, tls = [], []
    with gen.streaming(B):
        for s in range(steps):
            o, tl, al = gen.step_with_taps(codes[s][:B])
            outs.append(o.cpu()); tls.append(tl.cpu())
    return torch.stack(outs), torch.stack(tls)


def test_batch_rows_independent_and_graph_equals_eager(gpu_lib, monkeypatch):
    """Size-independent properties at the benchmark batch (B=32, 7B layer shapes, 2 layers): a session's tokens and
    logits do not depend on its neighbours (row 0 is bit-identical when the other 31 sessions carry different audio),
    and hipGraph replay == eager launches."""
    cfg = LMConfig(num_layers=2, context=64)
    sd = random_lm_state_dict(cfg, seed=11, device=DEV)
    g = torch.Generator().manual_seed(2)
    steps = 4
    codes = torch.randint(0, cfg.card, (steps, 32, 8, 1), generator=g).to(DEV)
    other = codes.clone()
    other[:, 1:] = torch.randint(0, cfg.card, (steps, 31, 8, 1), generator=g).to(DEV)
    o32, t32 = _greedy_run(cfg, sd, 32, codes, steps)
    o2, t2 = _greedy_run(cfg, sd, 32, other, steps)
    assert torch.equal(o32[:, :1], o2[:, :1]) and torch.equal(t32[:, :1], t2[:, :1])
    assert not torch.equal(t32[:, 1:], t2[:, 1:])
    monkeypatch.setenv("MMI_NO_GRAPH", "1")
    oe, te = _greedy_run(cfg, sd, 32, codes, steps)
    assert torch.equal(oe, o32) and torch.equal(te, t32)


def test_depformer_attention_inside_out_proj_is_bit_identical_to_its_own_launch(gpu_lib, monkeypatch, B=1):
    """One session, bf16 weights: the depth transformer's attention runs inside its out_proj (k_dep_attn_out_proj: the same
    per-(session, head) function, the same operand values).  Tokens and text logits equal those of the two-launch form
    (MMI_NO_DEP_ATTN_FUSION=1) bit for bit, at the 7B layer shapes; the audio tokens are what the depth transformer produced."""
    cfg = LMConfig(num_layers=2, context=64)
    sd = random_lm_state_dict(cfg, seed=13, device=DEV)
    steps = 5
    codes = torch.randint(0, cfg.card, (steps, B, 8, 1), generator=torch.Generator().manual_seed(4)).to(DEV)
    of, tf = _greedy_run(cfg, sd, B, codes, steps)
    monkeypatch.setenv("MMI_NO_DEP_ATTN_FUSION", "1")
    ou, tu = _greedy_run(cfg, sd, B, codes, steps)
    assert torch.equal(of, ou) and torch.equal(tf, tu)


def test_rng_sampling_statistics(gpu_lib):
    lm_cases.rng_sampling_statistics(DEV, None)


@pytest.mark.parametrize("name", ["e", "f"])
def test_cross_attention_conditioning_matches_reference_golden(gpu_lib, name):
    """A model with cross-attention layers (transformer.py:727-732, 779-797) fed by ConditionFuser.get_cross (base.py:392-409):
    `e` cross + sum conditions, `f` guidance with two concatenated cross tensors and the sinusoidal position embedding."""
    lm_cases.check_cross_engine(DEV, None, name)


@pytest.mark.parametrize("B,Tc", [(3, 7), (20, 37)])
def test_cross_attention_at_full_width_matches_oracle(gpu_lib, B, Tc):
    """Cross-attention layers at the 7B layer shapes (head dim 128: 16 chunk lanes x 4 position slots per pass; the condition's
    keys / values projected through the weight-streaming GEMM in batches) against the oracle, both batch tilings."""
    from dataclasses import replace
    lm_cases.cross_vs_oracle(DEV, None, replace(LMConfig(num_layers=2, context=64), cross_attention=True), B=B, S=2, Tc=Tc, seed=77 + B)


def test_attention_program_switch_is_a_graph_launch_not_a_capture(gpu_lib, monkeypatch):
    """with the ring split over workgroups (< 4 sessions) a stream moves from the short-ring step program to the
    deep-ring one after ~768 steps.  Both programs are captured and instantiated at the stream's FIRST step (mmi_lm_stat 1 == 0b11),
    so the step at the switch costs what its neighbours cost (host wall time with a synchronisation per step: no spike), and a
    snapshot restored at a shallow depth goes back to the short-ring program (mmi_lm_stat 2 = the restored offsets' maximum)."""
    import time
    from dataclasses import replace
    monkeypatch.setenv("MMI_ATTN_SOLO", "40")
    cfg = replace(tiny_lm_config(), context=600)
    sd = random_lm_state_dict(cfg, seed=45)
    lm = LMModel(sd, cfg, device=DEV, max_batch=1)
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    rng = np.random.default_rng(45)
    ms = []
    with gen.streaming(1):
        for s in range(80):
            codes = torch.from_numpy(rng.integers(0, cfg.card, (1, 8, 1))).to(DEV)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gen.step(codes)
            torch.cuda.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
            if s == 0:
                assert gen._lib.mmi_lm_stat(lm._handle, 1) == 3, "both step programs must exist after the first step"
            if s == 10:
                snap = gen.get_streaming_state()
        assert gen._lib.mmi_lm_stat(lm._handle, 2) >= 79
        gen.set_streaming_state(snap)
        assert gen._lib.mmi_lm_stat(lm._handle, 2) == 11       # the snapshot's offset, not "assume deep"
    steady = float(np.median(ms[5:35]))
    at_switch = max(ms[36:46])
    print(f"[switch] steady step {steady:.3f} ms, worst step around the switch {at_switch:.3f} ms, first step {ms[0]:.1f} ms")
    # (the deterministic half of the check is the readiness mask above; the timing half leaves room for a host hiccup - capturing and
    # instantiating a step program costs tens of milliseconds)
    assert at_switch <= 3.0 * steady + 3.0, f"the step at the program switch is a latency spike: {at_switch:.3f} ms against {steady:.3f}"


def test_benchmark_batch_step_is_bit_reproducible_between_streams(gpu_lib):
    """32 sessions, bf16, 7B layer widths (the benchmark's kernels): repeated streams on one handle equal the first bit for bit."""
    lm_cases.reproducible_between_streams(DEV, None, LMConfig(num_layers=2, context=64), B=32, quantize=False, seed=15, repeats=3)


@pytest.mark.parametrize("B", [3, 12])
def test_16_row_tile_step_is_bit_reproducible_between_streams(gpu_lib, B):
    """<= 16 sessions (the 16-row MFMA tile: C3's kernels - k_gemm_xp<16, ..>, k_gemm_xp_norm<16, ..>, k_gemm_xp_once<16, ..>, the
    ring split over workgroups at 3 sessions), 7B layer widths: repeated streams on one handle equal the first bit for bit."""
    lm_cases.reproducible_between_streams(DEV, None, LMConfig(num_layers=2, context=64), B=B, quantize=False, seed=21 + B, repeats=3)


def test_guided_cross_attention_step_is_bit_reproducible_between_streams(gpu_lib):
    """A cross-attention model under classifier-free guidance at the 7B layer widths (9 sessions = 18 model rows: the 32-row tile,
    norm_cross / query GEMM / k_lm_cross_attn / out_proj, k_cfg_mix at all nine sampling sites): repeated streams are bit-identical."""
    from dataclasses import replace
    cfg = replace(LMConfig(num_layers=2, context=64), cross_attention=True)
    lm_cases.reproducible_between_streams(DEV, None, cfg, B=9, quantize=False, seed=33, repeats=3, guided_cross=(7, 2.0))
