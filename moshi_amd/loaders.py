"""Checkpoint loading for the engine: same function names, arguments and file formats as the reference's
`moshi.models.loaders` (moshi/moshi/models/loaders.py:145-446), minus the Hugging Face download
(there is no network path in the engine: `hf://` names must already be local files).

    mimi = loaders.get_mimi("tokenizer-e351c8d8-checkpoint125.safetensors", device="cuda", num_codebooks=8)
    lm   = loaders.get_moshi_lm("model.safetensors", device="cuda", max_batch=32)              # Moshi-7B `_lm_kwargs`
    info = loaders.CheckpointInfo.from_local("/models/moshiko")      # a directory holding config.json + the files it names
    loaders.export_quantized("model.safetensors", "model.q8.safetensors")                      # scripts/export_quantized.py

The reference builds torch modules and `load_state_dict`s into them; here the state dict goes straight to the engine, which
repacks it into MFMA fragment order on the GPU (`mmi_mimi_create` / `mmi_lm_create`).  Configuration dictionaries are the
reference's own (`_mimi_config`, `_lm_kwargs`, or the `config.json` of a released checkpoint); options the kernels fix are
checked, not silently ignored.
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, Optional

import torch

from .config import LMConfig, MimiConfig
from .lm import ConditionFuser, LMModel
from .mimi import MimiModel


# Names the reference's callers import from this module (server.py / run_inference.py argparse defaults, scripts): the released
# repository and the file names inside it (loaders.py:28-35)
SAMPLE_RATE = 24000
FRAME_RATE = 12.5
TEXT_TOKENIZER_NAME = "tokenizer_spm_32k_3.model"
MOSHI_NAME = "model.safetensors"
MOSHI_Q8_NAME = "model.q8.safetensors"
MIMI_NAME = "tokenizer-e351c8d8-checkpoint125.safetensors"
DEFAULT_REPO = "kyutai/moshiko-pytorch-bf16"


def hf_get(filename: str | Path, hf_repo: str | None = None, check_local_file_exists: bool = False,
           revision: str | None = None) -> Path:
    """loaders.py:122-142 without the download: a Path, a `file://` name or a plain name is returned as a local path exactly as
    the reference does; a name that the reference would fetch (`hf://...`, or a bare name with `hf_repo` that is not already a
    local file) raises - the engine has no network path (section 9 of DESIGN.md)."""
    if isinstance(filename, Path):
        return filename
    if filename.startswith("file://"):
        return Path(filename[len("file://"):])
    if filename.startswith("hf://") or (hf_repo is not None and not (check_local_file_exists and Path(filename).exists())):
        raise RuntimeError(f"no network path in the engine: fetch {filename!r} (repository {hf_repo!r}) beforehand and pass the local file, "
                           "or use CheckpointInfo.from_local(dir)")
    return Path(filename)


def _is_safetensors(path: Path | str) -> bool:          # loaders.py:319-320
    return Path(path).suffix in (".safetensors", ".sft", ".sfts")


def _load_state(filename: Path | str, torch_key: Optional[tuple] = None) -> Dict[str, torch.Tensor]:
    if _is_safetensors(filename):
        from safetensors.torch import load_file
        return load_file(str(filename), device="cpu")
    pkg = torch.load(str(filename), "cpu")               # loaders.py:359-360, 424-426
    for k in torch_key or ():
        pkg = pkg[k]
    return pkg


def _require(cfg: dict, key: str, allowed, what: str):
    if key in cfg and cfg[key] not in allowed:
        raise ValueError(f"{what}: `{key}` = {cfg[key]!r} is not implemented by the engine (supported: {allowed})")


def mimi_config_from_dict(mimi_config: Optional[dict]) -> MimiConfig:
    """`_mimi_config`-shaped dict (loaders.py:38-88) -> MimiConfig, checking the options the kernels fix."""
    if mimi_config is None:
        return MimiConfig()
    sea, qz, tr = mimi_config["seanet"], mimi_config["quantizer"], mimi_config["transformer"]
    for key, allowed in (("causal", (True,)), ("n_residual_layers", (1,)), ("activation", ("ELU",)), ("norm", ("none",)),
                         ("pad_mode", ("constant",)), ("true_skip", (True,)), ("disable_norm_outer_blocks", (0,))):
        _require(sea, key, allowed, "Mimi SEANet")
    for key, allowed in (("causal", (True,)), ("conv_layout", (True,)), ("gating", ("none",)), ("norm", ("layer_norm",)),
                         ("positional_embedding", ("rope",))):
        _require(tr, key, allowed, "Mimi transformer")
    return MimiConfig(
        sample_rate=mimi_config["sample_rate"], frame_rate=mimi_config["frame_rate"], channels=mimi_config["channels"],
        dimension=sea["dimension"], n_filters=sea["n_filters"], ratios=list(sea["ratios"]), kernel_size=sea["kernel_size"],
        last_kernel_size=sea["last_kernel_size"], residual_kernel_size=sea["residual_kernel_size"], compress=sea["compress"],
        tr_d_model=tr["d_model"], tr_num_heads=tr["num_heads"], tr_num_layers=tr["num_layers"],
        tr_dim_feedforward=tr["dim_feedforward"], tr_context=tr["context"], tr_max_period=float(tr["max_period"]),
        tr_layer_scale=tr.get("layer_scale", 0.01), q_dimension=qz["dimension"], q_bins=qz["bins"], q_n_q=qz["n_q"])


def lm_config_from_kwargs(lm_kwargs: Optional[dict]) -> LMConfig:
    """`_lm_kwargs`-shaped dict (loaders.py:90-119, or the `config.json` of a released model) -> LMConfig."""
    if lm_kwargs is None:
        return LMConfig()
    kw = dict(lm_kwargs)
    for key, allowed in (("causal", (True,)), ("layer_scale", (None,)), ("gating", ("silu",)), ("norm", ("rms_norm_f32",)),
                         ("positional_embedding", ("rope",)), ("depformer_layer_scale", (None,)), ("depformer_multi_linear", (True,)),
                         ("depformer_gating", ("silu",)), ("depformer_pos_emb", ("none",)), ("depformer_weights_per_step", (True,)),
                         ("demux_second_text_stream", (False,)), ("demux_second_stream", (False,)),
                         ("depformer_low_rank_embeddings", (None,)), ("text_card_out", (None, kw.get("text_card")))):
        _require(kw, key, allowed, "Moshi LM")
    if "depformer_context" in kw and kw["depformer_context"] < kw["dep_q"]:
        raise ValueError("depformer_context must cover the dep_q micro-steps")
    if kw.get("depformer_weights_per_step_schedule") is not None:
        raise ValueError("depformer_weights_per_step_schedule is not implemented by the engine")
    return LMConfig(
        dim=kw["dim"], num_heads=kw["num_heads"], num_layers=kw["num_layers"], hidden_scale=kw.get("hidden_scale", 4.125),
        context=kw["context"], max_period=float(kw.get("max_period", 10000)), n_q=kw["n_q"], dep_q=kw["dep_q"], card=kw["card"],
        text_card=kw["text_card"], existing_text_padding_id=kw.get("existing_text_padding_id", 3),
        existing_text_end_padding_id=kw.get("existing_text_end_padding_id", 0),
        depformer_dim=kw["depformer_dim"], depformer_dim_feedforward=int(kw["depformer_dim_feedforward"]),
        depformer_num_heads=kw["depformer_num_heads"], depformer_num_layers=kw["depformer_num_layers"],
        delays=list(kw["delays"]), extra_heads_num_heads=kw.get("extra_heads_num_heads", 0),
        extra_heads_dim=kw.get("extra_heads_dim", 6), cross_attention=bool(kw.get("cross_attention", False)))


def get_mimi(filename: str | Path | None, mimi_config: dict | None = None, device: torch.device | str = "cuda",
             num_codebooks: int = 8, max_batch: int = 64, lib=None) -> MimiModel:
    """loaders.get_mimi (loaders.py:323-363).  `filename=None` gives seeded random weights of the architecture (the
    reference returns an uninitialised model there)."""
    cfg = mimi_config_from_dict(mimi_config)
    if filename is None:
        from .weights import random_mimi_state_dict
        state = random_mimi_state_dict(cfg, seed=0)
    else:
        state = _load_state(filename, ("model",))
    return MimiModel(state, cfg, device=device, max_batch=max_batch, num_codebooks=num_codebooks, lib=lib)


def get_condition_fuser(cfg: dict) -> ConditionFuser:    # loaders.py:476-483
    fuser_cfg = cfg["fuser"]
    return ConditionFuser({k: fuser_cfg.get(k, []) for k in ("sum", "cross", "prepend")},
                          cross_attention_pos_emb=bool(fuser_cfg.get("cross_attention_pos_emb", False)),
                          cross_attention_pos_emb_scale=float(fuser_cfg.get("cross_attention_pos_emb_scale", 1.0)))


def get_moshi_lm(filename: str | Path | None, lm_kwargs: Optional[Dict[str, Any]] = None, device: torch.device | str = "cuda",
                 dtype: torch.dtype = torch.bfloat16, lora_weights: str | Path | None = None, fuse_lora: bool = False,
                 lm_kwargs_overrides: Optional[dict] = None, max_batch: int = 32, lib=None, state_patch=None,
                 quantize: Optional[bool | str] = None) -> LMModel:
    """loaders.get_moshi_lm (loaders.py:366-446): bf16 weights; `quantize` in the config (a `.q8` checkpoint carrying int8
    `weight` + `weight_scb`, or fp8 `weight` + `weight_scale`) is taken from the tensors themselves."""
    assert dtype == torch.bfloat16, "the engine computes the LM in bf16 (fp32 accumulation), like the reference's default"
    kw = dict(lm_kwargs) if lm_kwargs is not None else None
    if kw is not None:
        kw.update(lm_kwargs_overrides or {})
        kw.pop("depformer_causal", None)                 # deprecated (loaders.py:394)
        lora = kw.pop("lora", False)
        kw.pop("lora_rank", None)
        lora_scaling = kw.pop("lora_scaling", 2.0)
        if lora_weights is not None and not lora:
            raise AssertionError("`lora` is False, but received some lora_weights to load.")     # loaders.py:442-445
        if lora and kw.get("quantize"):
            raise AssertionError("LoRA and quantization are incompatible for now.")              # loaders.py:429-431
        if "conditioners" in kw and kw["conditioners"]:
            raise NotImplementedError("condition providers (text / tensor conditioners) run outside the engine: pass their "
                                      "output as LMGen(condition_tensors=...)")
    if kw is None:
        lora, lora_scaling = False, 2.0
        if lora_weights is not None:
            raise AssertionError("`lora` is False, but received some lora_weights to load.")
    fuser = get_condition_fuser(kw) if kw is not None and kw.get("fuser") is not None else None
    cfg = lm_config_from_kwargs(kw)
    if quantize is None:                                # True / "int8": the reference's int8 storage; "fp8": the fp8 MFMA path
        quantize = bool(kw.get("quantize", False)) if kw is not None else False
    if quantize == "int8":
        quantize = True
    if filename is None:
        from .weights import random_lm_state_dict
        state = random_lm_state_dict(cfg, seed=0)
    else:
        state = _load_state(filename, ("fsdp_best_state", "model"))
        state = {k: v for k, v in state.items() if not (k.startswith("condition_provider.") or k.startswith("fuser."))}
    if kw is not None and lora and lora_weights is not None:
        # loaders.py:486-516: the adapter is merged into the base weights (the engine has no unfused LoRA path)
        assert _is_safetensors(lora_weights), "LoRA weights must be a safetensors file."
        from .weights import fuse_lora_state_dict, normalize_lm_state_dict
        state = fuse_lora_state_dict(normalize_lm_state_dict(state, cfg), _load_state(lora_weights), float(lora_scaling))
    if state_patch is not None:
        state_patch(state)
    already = any(v.dtype in (torch.int8, torch.float8_e4m3fn) for v in state.values())
    return LMModel(state, cfg, device=device, max_batch=max_batch, lib=lib, quantize=False if already else quantize, fuser=fuser)


def export_quantized(src: str | Path, dst: str | Path, fmt: str = "int8", lm_kwargs: Optional[dict] = None) -> Dict[str, int]:
    """scripts/export_quantized.py:38-64 without the hub: read a bf16 Moshi checkpoint, convert the linears
    (`replace_linear_with_qlinear`: temporal + depth transformers, depformer_in, linears, text_linear) to the reference's int8
    storage (`weight` int8 + `weight_scb`) or to fp8 (`weight` e4m3fn + `weight_scale`), and write a safetensors file."""
    from safetensors.torch import save_file

    from .weights import normalize_lm_state_dict, quantize_lm_state_dict, quantize_lm_state_dict_fp8
    state = _load_state(src, ("fsdp_best_state", "model"))
    # released checkpoints fuse the per-step attention projections (`self_attn.in_proj_weight`): split them first, as the
    # reference's load hook does before `replace_linear_with_qlinear` sees the modules (transformer.py:422-446)
    cfg = lm_config_from_kwargs(lm_kwargs)
    if lm_kwargs is None:                               # no config: the fused depformer projections tell the number of steps
        for key, val in state.items():
            if key.startswith("depformer.") and key.endswith(".self_attn.in_proj_weight"):
                dd = val.shape[1]
                assert val.shape[0] % (3 * dd) == 0, f"{key}: unexpected fused shape {tuple(val.shape)}"
                mult = val.shape[0] // (3 * dd)
                if mult != cfg.dep_q:
                    raise ValueError(f"{src}: the checkpoint fuses {mult} depformer steps but no lm_kwargs were given "
                                     f"(default dep_q = {cfg.dep_q}): pass the model's lm_kwargs")
                break
    state = normalize_lm_state_dict(state, cfg)
    if cfg.cross_attention or any(".cross_attention." in k for k in state):
        raise NotImplementedError("quantised export of a model with cross-attention layers is not supported (the engine runs them bf16 only)")
    if fmt == "int8":
        out = quantize_lm_state_dict(state)
    elif fmt == "fp8":
        out = quantize_lm_state_dict_fp8(state)
    else:
        raise ValueError("fmt must be 'int8' or 'fp8'")
    save_file({k: v.contiguous() for k, v in out.items()}, str(dst))
    return {"tensors": len(out), "quantized": sum(v.dtype in (torch.int8, torch.float8_e4m3fn) for v in out.values()),
            "bytes": sum(v.numel() * v.element_size() for v in out.values())}


@dataclass
class CheckpointInfo:
    """The local half of the reference's `CheckpointInfo` (loaders.py:145-317): paths of the sub-models + their configs."""
    moshi_weights: Path
    mimi_weights: Path
    tokenizer: Optional[Path] = None
    lm_config: Optional[dict] = None
    raw_config: Optional[dict] = None
    mimi_config: Optional[dict] = None
    model_type: str = "moshi"
    lora_weights: Optional[Path] = None
    lm_gen_config: dict = field(default_factory=dict)
    tts_config: dict = field(default_factory=dict)
    stt_config: dict = field(default_factory=dict)
    model_id: dict = field(default_factory=dict)

    @staticmethod
    def from_hf_repo(hf_repo: str = DEFAULT_REPO, moshi_weights=None, mimi_weights=None, tokenizer=None, config_path=None,
                     mimi_config_path=None, lora_weights=None, revision=None):
        raise RuntimeError("the engine has no network path: download the repository and use CheckpointInfo.from_local(dir)")

    @staticmethod
    def from_local(path: str | Path, moshi_weights=None, mimi_weights=None, tokenizer=None, lora_weights=None,
                   mimi_config_path=None) -> "CheckpointInfo":
        """`path`: a directory with `config.json` (the released layout, loaders.py:181-280) or the config file itself.
        config.json keys: `moshi_name`, `mimi_name`, `tokenizer_name`, optional `model_type`, `lm_gen_config`, `mimi_config` /
        `mimi_config_name`, `lora_name`, `tts_config`, `stt_config`, `model_id`; everything else is the LM's kwargs.  A
        fine-tune's adapter (`lora_name`, or `lora_weights=`) is resolved here and merged at load (loaders.py:229,260-265,305)."""
        path = Path(path)
        cfg_file = path / "config.json" if path.is_dir() else path
        root = cfg_file.parent
        raw = json.loads(cfg_file.read_text())
        lm_config = dict(raw)
        names = {k: lm_config.pop(k, None) for k in ("moshi_name", "mimi_name", "tokenizer_name")}
        model_type = lm_config.pop("model_type", "moshi")
        lm_gen_config = lm_config.pop("lm_gen_config", {})
        mimi_config = lm_config.pop("mimi_config", None)
        mimi_config_name = lm_config.pop("mimi_config_name", None)
        lora_name = lm_config.pop("lora_name", None)
        tts_config = lm_config.pop("tts_config", {})
        stt_config = lm_config.pop("stt_config", {})
        model_id = lm_config.pop("model_id", {})

        def local(given, name):
            if given is not None:
                return Path(given)
            if name is None:
                return None
            if str(name).startswith("hf://"):
                name = str(name).rsplit("/", 1)[-1]     # the file must already sit next to config.json
            return root / name
        mimi_cfg_file = local(mimi_config_path, mimi_config_name)            # loaders.py:251-258
        if mimi_cfg_file is not None:
            if not mimi_cfg_file.exists():
                raise FileNotFoundError(f"mimi config {mimi_cfg_file} named by {cfg_file} is missing")
            mimi_config = json.loads(mimi_cfg_file.read_text())
        lora_file = local(lora_weights, lora_name)                          # loaders.py:260-265
        if lm_config.get("lora", False):
            if lora_file is None or not lora_file.exists():
                raise FileNotFoundError(f"{cfg_file} describes a LoRA fine-tune (`lora`: true) but no adapter file was found "
                                        f"({lora_file}): refusing to run the base weights under that name")
        elif lora_file is not None and lora_weights is None:
            lora_file = None                                                 # `lora_name` without `lora`: nothing to merge
        return CheckpointInfo(local(moshi_weights, names["moshi_name"]), local(mimi_weights, names["mimi_name"]),
                              local(tokenizer, names["tokenizer_name"]), lm_config=lm_config, raw_config=raw,
                              mimi_config=mimi_config, model_type=model_type, lora_weights=lora_file,
                              lm_gen_config=lm_gen_config, tts_config=tts_config, stt_config=stt_config, model_id=model_id)

    def get_mimi(self, device: torch.device | str = "cuda", **kwargs) -> MimiModel:      # loaders.py:282-291
        n = 8 if self.lm_config is None else max(self.lm_config["dep_q"], self.lm_config["n_q"] - self.lm_config["dep_q"])
        return get_mimi(self.mimi_weights, self.mimi_config, device=device, num_codebooks=n, **kwargs)

    def get_text_tokenizer(self):                                                         # loaders.py:315-316
        import sentencepiece
        if self.tokenizer is None:
            raise FileNotFoundError("this checkpoint names no tokenizer (`tokenizer_name` in config.json)")
        return sentencepiece.SentencePieceProcessor(str(self.tokenizer))

    def get_moshi(self, device: torch.device | str = "cuda", dtype: torch.dtype = torch.bfloat16, load_weight: bool = True,
                  **kwargs) -> LMModel:
        def hibiki(state):   # loaders.py:308-312: an early EOS (2) is read as PAD (3)
            w = state["text_emb.weight"].clone()
            w[2] = w[3]
            state["text_emb.weight"] = w
        kwargs.setdefault("lora_weights", self.lora_weights)                                         # loaders.py:305
        return get_moshi_lm(self.moshi_weights if load_weight else None, lm_kwargs=self.lm_config, device=device, dtype=dtype,
                            state_patch=hibiki if self.model_type == "hibiki" else None, **kwargs)   # loaders.py:293-313
