"""Checkpoint key normalisation: the fused / legacy names released checkpoints use map onto the per-step names the
engine (and the current reference) expect - modules/transformer.py:422-446, quantization/core_vq.py:162-176."""
import torch

from moshi_amd.config import tiny_lm_config, tiny_mimi_config
from moshi_amd.weights import (normalize_lm_state_dict, normalize_mimi_state_dict, random_lm_state_dict,
                               random_mimi_state_dict)


def test_lm_fused_projections_are_split_per_step():
    cfg = tiny_lm_config()
    sd = random_lm_state_dict(cfg, seed=1)
    fused = dict(sd)
    for l in range(cfg.depformer_num_layers):
        p = f"depformer.layers.{l}.self_attn."
        fused[p + "in_proj_weight"] = torch.cat([fused.pop(p + f"in_projs.{k}.weight") for k in range(cfg.dep_q)], 0)
        fused[p + "out_proj.weight"] = torch.cat([fused.pop(p + f"out_projs.{k}.weight") for k in range(cfg.dep_q)], 0)
    for l in range(cfg.num_layers):
        p = f"transformer.layers.{l}.self_attn."
        fused[p + "in_proj_weight"] = fused.pop(p + "in_projs.0.weight")
        fused[p + "out_proj.weight"] = fused.pop(p + "out_projs.0.weight")
    back = normalize_lm_state_dict(fused, cfg)
    assert set(back) == set(sd)
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert normalize_lm_state_dict(sd, cfg).keys() == sd.keys()          # canonical names pass through


def test_mimi_legacy_codebook_names():
    cfg = tiny_mimi_config()
    sd = random_mimi_state_dict(cfg, seed=2)
    legacy = {}
    for k, v in sd.items():
        k = k.replace("._codebook.cluster_usage", "._codebook.cluster_size").replace("._codebook.embedding_sum", "._codebook.embed_sum")
        k = k.replace("._codebook._initialized", "._codebook.inited")
        k = k.replace("self_attn.in_projs.0.weight", "self_attn.in_proj_weight").replace("self_attn.out_projs.0.weight", "self_attn.out_proj.weight")
        legacy[k] = v
    assert set(legacy) != set(sd)
    back = normalize_mimi_state_dict(legacy)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k]) for k in sd)


def test_fused_quantised_and_lora_keys_are_split_with_their_weights():
    """The reference's load hook also splits the `_scb` row scales of a quantised checkpoint and fused LoRA factors
    (transformer.py:423-446); so does normalize_lm_state_dict."""
    import torch
    from moshi_amd.config import tiny_lm_config
    cfg = tiny_lm_config()
    dd, q = cfg.depformer_dim, cfg.dep_q
    p = "depformer.layers.0.self_attn."
    sd = {p + "in_proj_weight": torch.zeros(q * 3 * dd, dd, dtype=torch.int8), p + "in_proj_weight_scb": torch.arange(q * 3 * dd, dtype=torch.float32),
          p + "out_proj.weight": torch.zeros(q * dd, dd, dtype=torch.int8), p + "out_proj.weight_scb": torch.arange(q * dd, dtype=torch.float32),
          p + "in_proj.lora_A.weight": torch.ones(q * 4, dd), p + "in_proj.lora_B.weight": torch.ones(q * 3 * dd, 4)}
    out = normalize_lm_state_dict(sd, cfg)
    for i in range(q):
        assert out[p + f"in_projs.{i}.weight"].shape == (3 * dd, dd)
        assert torch.equal(out[p + f"in_projs.{i}.weight_scb"], torch.arange(i * 3 * dd, (i + 1) * 3 * dd, dtype=torch.float32))
        assert torch.equal(out[p + f"out_projs.{i}.weight_scb"], torch.arange(i * dd, (i + 1) * dd, dtype=torch.float32))
        assert out[p + f"in_projs.{i}.lora_A.weight"].shape == (4, dd) and out[p + f"in_projs.{i}.lora_B.weight"].shape == (3 * dd, 4)
    assert not any(k.endswith("in_proj_weight") or k.endswith("in_proj_weight_scb") or ".in_proj.lora" in k for k in out)
