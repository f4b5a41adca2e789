"""Checkpoint loading on the real device: a released-style directory (config.json + safetensors with the
reference's keys) -> CheckpointInfo -> MimiModel / LMModel on cuda:0 through the product library, and the quantised exports
through the same door.  The models built straight from the state dict are pinned on the oracle / the reference's golden vectors
by test_b_lm_gpu.py and test_a_mimi_gpu.py; here the loaders must hand the engine exactly the same weights."""
import json
from dataclasses import replace

import numpy as np
import pytest
import torch
from safetensors.torch import load_file, save_file

from moshi_amd import loaders
from moshi_amd.config import tiny_lm_config, tiny_mimi_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.mimi import MimiModel
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def tiny_lm_kwargs():
    kw = tiny_lm_config().reference_kwargs()
    kw["depformer_causal"] = True
    return kw


def greedy_tokens(lm, steps=5, B=3):
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    rng = np.random.default_rng(0)
    outs = []
    with gen.streaming(B):
        for _ in range(steps):
            outs.append(gen.step(torch.from_numpy(rng.integers(0, lm.card, (B, 8, 1))).to(DEV)).cpu().numpy())
    return np.stack(outs)


def test_released_style_directory_loads_onto_the_gpu(gpu_lib, tmp_path):
    lcfg = tiny_lm_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.dep_q)
    lsd, msd = random_lm_state_dict(lcfg, seed=3), random_mimi_state_dict(mcfg, seed=4)
    save_file(lsd, str(tmp_path / "model.safetensors"))
    save_file(msd, str(tmp_path / "mimi.safetensors"))
    conf = {**tiny_lm_kwargs(), "moshi_name": "model.safetensors", "mimi_name": "hf://kyutai/some-repo/mimi.safetensors",
            "tokenizer_name": "tokenizer.model", "model_type": "moshi", "lm_gen_config": {"temp": 0.7},
            "mimi_config": mcfg.reference_kwargs()}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    info = loaders.CheckpointInfo.from_local(tmp_path)
    mimi = info.get_mimi(device=DEV, max_batch=3)
    lm = info.get_moshi(device=DEV, max_batch=3)
    assert lm.config == lcfg and mimi.num_codebooks == 8
    # the language model: same greedy dialogue as the model built from the state dict
    assert np.array_equal(greedy_tokens(lm), greedy_tokens(LMModel(lsd, lcfg, device=DEV, max_batch=3)))
    # the codec: same codes and PCM as the model built from the state dict
    ref = MimiModel(msd, mcfg, device=DEV, max_batch=3, num_codebooks=8)
    x = torch.from_numpy((0.3 * np.random.default_rng(1).standard_normal((3, 1, mcfg.frame_size * 4))).astype(np.float32)).to(DEV)
    with mimi.streaming(3), ref.streaming(3):
        c1, c2 = mimi.encode(x), ref.encode(x)
        assert torch.equal(c1, c2)
        assert torch.equal(mimi.decode(c1), ref.decode(c2))
    # one duplex frame through the loaded pair: PCM -> codes -> LMGen.step -> PCM
    gen = LMGen(lm, use_sampling=False)
    played = 0
    with mimi.streaming(3), gen.streaming(3):
        for f in range(4):
            codes = mimi.encode(x[:, :, f * mcfg.frame_size:(f + 1) * mcfg.frame_size])
            out = gen.step(codes)
            if out is None:          # the acoustic delay (lm.py:779-783)
                continue
            pcm = mimi.decode(out[:, 1:])
            assert pcm.shape == (3, 1, mcfg.frame_size) and bool(torch.isfinite(pcm).all())
            played += 1
    assert played >= 1


@pytest.mark.parametrize("fmt", ["int8", "fp8"])
def test_quantised_export_loads_onto_the_gpu(gpu_lib, tmp_path, fmt):
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=11)
    save_file(sd, str(tmp_path / "model.safetensors"))
    loaders.export_quantized(tmp_path / "model.safetensors", tmp_path / f"model.{fmt}.safetensors", fmt, tiny_lm_kwargs())
    stored = load_file(str(tmp_path / f"model.{fmt}.safetensors"))
    assert any(v.dtype in (torch.int8, torch.float8_e4m3fn) for v in stored.values())
    lm = loaders.get_moshi_lm(tmp_path / f"model.{fmt}.safetensors", {**tiny_lm_kwargs(), "quantize": True}, device=DEV, max_batch=3)
    want = greedy_tokens(LMModel(sd, cfg, device=DEV, max_batch=3, quantize=True if fmt == "int8" else "fp8"))
    assert np.array_equal(greedy_tokens(lm), want)
