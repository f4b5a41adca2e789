"""Offline inference driver (moshi_amd/run_inference.py, the reference's run_inference.py) on the CPU kernel simulator."""
import json
from dataclasses import replace

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from moshi_amd import loaders, run_inference
from moshi_amd.config import tiny_lm_config, tiny_mimi_config
from moshi_amd.lm import LMGen
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict


class StubTokenizer:
    def __init__(self, eos=-1):
        self._eos = eos

    def eos_id(self):
        return self._eos

    def id_to_piece(self, i):
        return f"▁t{i}"


def make_dir(tmp_path, model_type="moshi"):
    lcfg = tiny_lm_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.dep_q)
    save_file(random_lm_state_dict(lcfg, seed=3), str(tmp_path / "model.safetensors"))
    save_file(random_mimi_state_dict(mcfg, seed=4), str(tmp_path / "mimi.safetensors"))
    conf = {**lcfg.reference_kwargs(), "moshi_name": "model.safetensors", "mimi_name": "mimi.safetensors", "model_type": model_type,
            "mimi_config": mcfg.reference_kwargs()}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    return loaders.CheckpointInfo.from_local(tmp_path), lcfg, mcfg


def test_run_matches_the_loop_written_by_hand(sim_lib, tmp_path):
    info, lcfg, mcfg = make_dir(tmp_path)
    B, n = 2, 5
    rng = np.random.default_rng(0)
    pcm = torch.from_numpy((0.3 * rng.standard_normal((B, 1, n * mcfg.frame_size + 17))).astype(np.float32))   # the ragged tail is dropped
    said = []
    st = run_inference.InferenceState(info, info.get_mimi("cpu", max_batch=B, lib=sim_lib), StubTokenizer(),
                                      info.get_moshi("cpu", max_batch=B, lib=sim_lib), B, device="cpu", use_sampling=False,
                                      on_token=said.append)
    out = st.run(pcm)
    # by hand (run_inference.py:160-171): the first frame is stepped twice
    mimi, lm = info.get_mimi("cpu", max_batch=B, lib=sim_lib), info.get_moshi("cpu", max_batch=B, lib=sim_lib)
    gen = LMGen(lm, use_sampling=False)
    mimi.streaming_forever(B); gen.streaming_forever(B)
    texts, pcms = [], []
    for f in range(n):
        codes = mimi.encode(pcm[..., f * mcfg.frame_size:(f + 1) * mcfg.frame_size])
        if f == 0:
            assert gen.step(codes) is None
        tokens = gen.step(codes)
        if tokens is None:
            continue
        texts.append(tokens[:, 0]); pcms.append(mimi.decode(tokens[:, 1:]))
    assert len(texts) == n - (lcfg.max_delay - 1) - 1 or len(texts) == n      # delays of 1: the doubled first step absorbs the delay
    for b in range(B):
        assert torch.equal(out[b][0], torch.cat([t[b] for t in texts]))
        assert torch.equal(out[b][1], torch.cat([p[b] for p in pcms], dim=1))
    assert len(said) == sum(int(t[0]) not in (0, 3) for t in texts)


def test_hibiki_feeds_end_of_stream_and_waits_for_eos(sim_lib, tmp_path):
    info, lcfg, mcfg = make_dir(tmp_path, "hibiki")
    assert info.model_type == "hibiki"
    n = 3
    pcm = torch.from_numpy((0.3 * np.random.default_rng(1).standard_normal((1, 1, n * mcfg.frame_size))).astype(np.float32))

    def state(eos):
        return run_inference.InferenceState(info, info.get_mimi("cpu", max_batch=1, lib=sim_lib), StubTokenizer(eos),
                                            info.get_moshi("cpu", max_batch=1, lib=sim_lib), 1, device="cpu", use_sampling=False,
                                            on_token=lambda t: None)
    probe = state(-1).run(pcm, max_steps=n + 4)[0][0]              # never sees EOS: bounded by max_steps
    assert len(probe) == n + 4
    # text tokens emitted after the end-of-stream code went in: make the 3rd of them the EOS id
    eos = int(probe[n + 1])
    first = next(i for i in range(n, len(probe)) if int(probe[i]) == eos)
    text, audio = state(eos).run(pcm, max_steps=50)[0]
    assert len(text) == first + 1 and int(text[-1]) == eos          # stops AT the first EOS after the input ended
    assert torch.equal(text, probe[:first + 1]) and audio.shape == (1, (first + 1) * mcfg.frame_size)


def test_wav_roundtrip_and_cli(sim_lib, tmp_path, monkeypatch):
    info, lcfg, mcfg = make_dir(tmp_path)
    x = 0.5 * np.sin(2 * np.pi * 50 * np.arange(4 * mcfg.frame_size) / mcfg.sample_rate).astype(np.float32)
    run_inference.write_wav(str(tmp_path / "in.wav"), x, mcfg.sample_rate)
    y = run_inference.read_wav(str(tmp_path / "in.wav"), mcfg.sample_rate)
    assert np.abs(x - y).max() < 1e-4 * 4
    with pytest.raises(ValueError, match="no resampler"):
        run_inference.read_wav(str(tmp_path / "in.wav"), 24000)
    # the CLI end to end on the simulator: route the loaders to it
    monkeypatch.setattr(loaders.CheckpointInfo, "get_mimi", lambda self, device="cuda", **kw: loaders.get_mimi(
        self.mimi_weights, self.mimi_config, device="cpu", num_codebooks=8, max_batch=kw.get("max_batch", 2), lib=sim_lib))
    monkeypatch.setattr(loaders.CheckpointInfo, "get_moshi", lambda self, device="cuda", **kw: loaders.get_moshi_lm(
        self.moshi_weights, self.lm_config, device="cpu", max_batch=kw.get("max_batch", 2), lib=sim_lib, quantize=kw.get("quantize")))
    run_inference.main(["--checkpoint-dir", str(tmp_path), "--batch-size", "1", "--device", "cpu", "--greedy",
                        str(tmp_path / "in.wav"), str(tmp_path / "out.wav")])
    out = run_inference.read_wav(str(tmp_path / "out.wav"), mcfg.sample_rate)
    assert out.shape[0] == 4 * mcfg.frame_size and np.isfinite(out).all()


def test_asr_style_model_prints_text_only(sim_lib, tmp_path):
    """dep_q = 0: every codebook of the codec is input, nothing is decoded, the text stream is what comes out."""
    from moshi_amd.config import tiny_stt_config
    lcfg = tiny_stt_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.n_q)
    save_file(random_lm_state_dict(lcfg, seed=3), str(tmp_path / "model.safetensors"))
    save_file(random_mimi_state_dict(mcfg, seed=4), str(tmp_path / "mimi.safetensors"))
    # model_type "stt": the input is padded by `audio_silence_prefix_seconds` on the left and `audio_delay_seconds` + 1 s on the
    # right (at 24 kHz, run_inference.py:121-127); a negative delay keeps the tiny model's padding to 1 + 2.5 frames
    F = mcfg.frame_size
    conf = {**lcfg.reference_kwargs(), "moshi_name": "model.safetensors", "mimi_name": "mimi.safetensors", "model_type": "stt",
            "mimi_config": mcfg.reference_kwargs(),
            "stt_config": {"audio_silence_prefix_seconds": F / 24000.0, "audio_delay_seconds": 2.5 * F / 24000.0 - 1.0}}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    info = loaders.CheckpointInfo.from_local(tmp_path)
    mimi = info.get_mimi("cpu", max_batch=1, lib=sim_lib)
    assert mimi.num_codebooks == lcfg.n_q                       # loaders.py:282-291: max(dep_q, n_q - dep_q)
    lm = info.get_moshi("cpu", max_batch=1, lib=sim_lib)
    assert lm.config.dep_q == 0 and lm.config.extra_heads_num_heads == 2
    said = []
    st = run_inference.InferenceState(info, mimi, StubTokenizer(), lm, 1, device="cpu", use_sampling=False, on_token=said.append)
    n = 6
    pcm = torch.from_numpy((0.3 * np.random.default_rng(2).standard_normal((1, 1, n * mcfg.frame_size))).astype(np.float32))
    (text, audio), = st.run(pcm)
    # 1 (left pad) + n + 2 (whole frames of the 2.5-frame right pad) frames + the doubled first step - the text delay
    assert audio.numel() == 0 and len(text) == (1 + n + 2) + 1 - lcfg.max_delay
    assert len(said) == sum(int(t) not in (0, 3) for t in text)
