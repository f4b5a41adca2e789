"""Checkpoint loading: safetensors files with the reference's keys and config dictionaries -> engine models,
quantised export.  Runs on the CPU kernel simulator."""
import json
from dataclasses import replace

import numpy as np
import pytest
import torch
from safetensors.torch import load_file, save_file

from moshi_amd import loaders
from moshi_amd.config import tiny_lm_config, tiny_mimi_config
from moshi_amd.lm import LMGen
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict


def tiny_lm_kwargs():
    kw = tiny_lm_config().reference_kwargs()          # the dict the reference's LMModel(**kw) takes (== config.json layout)
    kw["depformer_causal"] = True                     # deprecated key still present in released configs (loaders.py:394)
    return kw


def greedy_tokens(lm, steps=4, B=2):
    gen = LMGen(lm, use_sampling=False, support_out_of_sync=True)
    rng = np.random.default_rng(0)
    outs = []
    with gen.streaming(B):
        for _ in range(steps):
            outs.append(gen.step(torch.from_numpy(rng.integers(0, lm.card, (B, 8, 1)))).numpy())
    return np.stack(outs)


def test_moshi_checkpoint_roundtrip_and_config_mapping(sim_lib, tmp_path):
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=8)
    save_file(sd, str(tmp_path / "model.safetensors"))
    lm = loaders.get_moshi_lm(tmp_path / "model.safetensors", tiny_lm_kwargs(), device="cpu", max_batch=2, lib=sim_lib)
    assert lm.config == cfg
    from moshi_amd.lm import LMModel
    ref = greedy_tokens(LMModel(sd, cfg, device="cpu", max_batch=2, lib=sim_lib))
    assert np.array_equal(greedy_tokens(lm), ref)
    # torch.save layout of training checkpoints (loaders.py:424-426)
    torch.save({"fsdp_best_state": {"model": sd}}, tmp_path / "model.pt")
    lm2 = loaders.get_moshi_lm(tmp_path / "model.pt", tiny_lm_kwargs(), device="cpu", max_batch=2, lib=sim_lib)
    assert np.array_equal(greedy_tokens(lm2), ref)


def test_unsupported_options_are_refused_not_ignored():
    kw = tiny_lm_kwargs()
    with pytest.raises(ValueError, match="norm"):
        loaders.lm_config_from_kwargs({**kw, "norm": "layer_norm"})
    with pytest.raises(ValueError, match="depformer_weights_per_step"):
        loaders.lm_config_from_kwargs({**kw, "depformer_weights_per_step": False})
    with pytest.raises(AssertionError, match="lora"):
        loaders.get_moshi_lm(None, kw, device="cpu", lora_weights="adapter.safetensors")
    mc = tiny_mimi_config().reference_kwargs()
    mc["seanet"]["pad_mode"] = "reflect"
    with pytest.raises(ValueError, match="pad_mode"):
        loaders.mimi_config_from_dict(mc)
    with pytest.raises(RuntimeError, match="no network"):
        loaders.CheckpointInfo.from_hf_repo("kyutai/moshiko-pytorch-bf16")


def test_default_configs_are_the_released_models():
    from moshi_amd.config import LMConfig, MimiConfig
    assert loaders.lm_config_from_kwargs(LMConfig().reference_kwargs()) == LMConfig()
    assert loaders.mimi_config_from_dict(MimiConfig().reference_kwargs()) == MimiConfig()
    assert loaders.lm_config_from_kwargs(None) == LMConfig() and loaders.mimi_config_from_dict(None) == MimiConfig()


@pytest.mark.parametrize("fmt,dtype,scale_key", [("int8", torch.int8, "_scb"), ("fp8", torch.float8_e4m3fn, "_scale")])
def test_export_quantized_writes_the_storage_the_engine_loads(sim_lib, tmp_path, fmt, dtype, scale_key):
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=9)
    save_file(sd, str(tmp_path / "model.safetensors"))
    # stored the way released checkpoints store the attention projections: fused over the depformer's steps
    fused = dict(sd)
    for l in range(cfg.depformer_num_layers):
        pre = f"depformer.layers.{l}.self_attn."
        fused[pre + "in_proj_weight"] = torch.cat([fused.pop(pre + f"in_projs.{k}.weight") for k in range(cfg.dep_q)])
        fused[pre + "out_proj.weight"] = torch.cat([fused.pop(pre + f"out_projs.{k}.weight") for k in range(cfg.dep_q)])
    save_file(fused, str(tmp_path / "model.safetensors"))
    info = loaders.export_quantized(tmp_path / "model.safetensors", tmp_path / f"model.{fmt}.safetensors", fmt, tiny_lm_kwargs())
    q = load_file(str(tmp_path / f"model.{fmt}.safetensors"))
    k = "transformer.layers.0.self_attn.in_projs.0.weight"
    assert q[k].dtype == dtype and q[k + scale_key].dtype == torch.float32 and q["emb.0.weight"].dtype == torch.bfloat16
    assert info["quantized"] == sum(v.dtype == dtype for v in q.values()) > 0
    lm = loaders.get_moshi_lm(tmp_path / f"model.{fmt}.safetensors", {**tiny_lm_kwargs(), "quantize": True}, device="cpu", max_batch=2,
                              lib=sim_lib)
    assert lm.quantized
    from moshi_amd.lm import LMModel
    ref = greedy_tokens(LMModel(sd, cfg, device="cpu", max_batch=2, lib=sim_lib, quantize=True if fmt == "int8" else "fp8"))
    assert np.array_equal(greedy_tokens(lm), ref)


def test_checkpoint_info_from_a_released_style_directory(sim_lib, tmp_path):
    lcfg = tiny_lm_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.dep_q)
    save_file(random_lm_state_dict(lcfg, seed=3), str(tmp_path / "model.safetensors"))
    save_file(random_mimi_state_dict(mcfg, seed=4), str(tmp_path / "mimi.safetensors"))
    conf = {**tiny_lm_kwargs(), "moshi_name": "model.safetensors", "mimi_name": "hf://kyutai/some-repo/mimi.safetensors",
            "tokenizer_name": "tokenizer.model", "model_type": "hibiki", "lm_gen_config": {"temp": 0.7},
            "mimi_config": mcfg.reference_kwargs(), "fuser": {"sum": ["description"], "cross": []}}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    info = loaders.CheckpointInfo.from_local(tmp_path)
    assert info.model_type == "hibiki" and info.lm_gen_config == {"temp": 0.7} and info.mimi_weights == tmp_path / "mimi.safetensors"
    mimi = info.get_mimi(device="cpu", max_batch=2, lib=sim_lib)
    assert mimi.num_codebooks == 8 and mimi.frame_size == mcfg.frame_size and mimi.cardinality == lcfg.card
    lm = info.get_moshi(device="cpu", max_batch=2, lib=sim_lib)
    assert lm.fuser is not None and lm.fuser.fuse2cond["sum"] == ["description"]
    x = torch.zeros(1, 1, mcfg.frame_size)
    with mimi.streaming(1):
        assert mimi.encode(x).shape == (1, 8, 1)


def test_lora_adapter_is_merged_like_the_reference(sim_lib, tmp_path):
    """loaders.get_lora_moshi with fuse_lora (modules/lora.py:26-43): W' = W + scaling * B @ A in the weights' dtype."""
    from moshi_amd.lm import LMModel
    from moshi_amd.weights import fuse_lora_state_dict
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=12)
    g = torch.Generator().manual_seed(0)
    rank, scaling = 4, 2.0
    lora = {}
    for stem in ("transformer.layers.0.self_attn.in_projs.0", "transformer.layers.1.gating.linear_out",
                 "depformer.layers.0.gating.3.linear_in", "text_linear", "linears.2"):
        w = sd[stem + ".weight"]
        lora[stem + ".lora_A.weight"] = (0.3 * torch.randn(rank, w.shape[1], generator=g)).to(torch.bfloat16)
        lora[stem + ".lora_B.weight"] = (0.3 * torch.randn(w.shape[0], rank, generator=g)).to(torch.bfloat16)
    merged = fuse_lora_state_dict(sd, lora, scaling)
    k = "text_linear.weight"
    want = sd[k] + scaling * (lora["text_linear.lora_B.weight"] @ lora["text_linear.lora_A.weight"])
    assert torch.equal(merged[k], want) and not torch.equal(merged[k], sd[k])
    assert torch.equal(merged["emb.0.weight"], sd["emb.0.weight"])
    with pytest.raises(RuntimeError, match="unexpected_keys"):
        fuse_lora_state_dict(sd, {"nope.lora_A.weight": lora["text_linear.lora_A.weight"], "nope.lora_B.weight": lora["text_linear.lora_B.weight"]}, 2.0)
    save_file(sd, str(tmp_path / "model.safetensors"))
    save_file(lora, str(tmp_path / "lora.safetensors"))
    lm = loaders.get_moshi_lm(tmp_path / "model.safetensors", {**tiny_lm_kwargs(), "lora": True, "lora_rank": rank, "lora_scaling": scaling},
                              device="cpu", max_batch=2, lib=sim_lib, lora_weights=tmp_path / "lora.safetensors", fuse_lora=True)
    assert np.array_equal(greedy_tokens(lm), greedy_tokens(LMModel(merged, cfg, device="cpu", max_batch=2, lib=sim_lib)))


def test_checkpoint_info_resolves_lora_and_mimi_config_names(sim_lib, tmp_path):
    """A fine-tune's config.json (`lora`: true, `lora_name`, `mimi_config_name`): the adapter is found and merged by get_moshi,
    the codec config is read from the named file, and a missing adapter is an error - never the base model run silently
    (loaders.py:226-265, 305)."""
    from moshi_amd.weights import fuse_lora_state_dict, normalize_lm_state_dict
    lcfg = tiny_lm_config()
    mcfg = replace(tiny_mimi_config(), q_bins=lcfg.card, q_n_q=lcfg.dep_q)
    sd = random_lm_state_dict(lcfg, seed=5)
    save_file(sd, str(tmp_path / "model.safetensors"))
    save_file(random_mimi_state_dict(mcfg, seed=6), str(tmp_path / "mimi.safetensors"))
    g = torch.Generator().manual_seed(1)
    w = sd["text_linear.weight"]
    lora = {"text_linear.lora_A.weight": (0.3 * torch.randn(4, w.shape[1], generator=g)).to(torch.bfloat16),
            "text_linear.lora_B.weight": (0.3 * torch.randn(w.shape[0], 4, generator=g)).to(torch.bfloat16)}
    (tmp_path / "mimi_config.json").write_text(json.dumps(mcfg.reference_kwargs()))
    conf = {**tiny_lm_kwargs(), "moshi_name": "model.safetensors", "mimi_name": "mimi.safetensors", "tokenizer_name": "tok.model",
            "lora": True, "lora_rank": 4, "lora_scaling": 2.0, "lora_name": "lora.safetensors", "mimi_config_name": "mimi_config.json",
            "stt_config": {"audio_delay_seconds": 0.5}, "model_id": {"sig": "abc"}}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    with pytest.raises(FileNotFoundError, match="adapter"):
        loaders.CheckpointInfo.from_local(tmp_path)
    save_file(lora, str(tmp_path / "lora.safetensors"))
    info = loaders.CheckpointInfo.from_local(tmp_path)
    assert info.lora_weights == tmp_path / "lora.safetensors" and info.stt_config == {"audio_delay_seconds": 0.5}
    assert info.mimi_config == mcfg.reference_kwargs() and "mimi_config_name" not in info.lm_config
    assert info.get_mimi(device="cpu", max_batch=2, lib=sim_lib).cardinality == lcfg.card
    from moshi_amd.lm import LMModel
    merged = fuse_lora_state_dict(normalize_lm_state_dict(dict(sd), lcfg), lora, 2.0)
    want = greedy_tokens(LMModel(merged, lcfg, device="cpu", max_batch=2, lib=sim_lib))
    base = greedy_tokens(LMModel(sd, lcfg, device="cpu", max_batch=2, lib=sim_lib))
    got = greedy_tokens(info.get_moshi(device="cpu", max_batch=2, lib=sim_lib))
    assert np.array_equal(got, want) and not np.array_equal(got, base)
