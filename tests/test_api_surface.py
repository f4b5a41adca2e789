"""The drop-in boundary: moshi_amd's MimiModel / LMModel / LMGen against the PUBLIC surface of the reference's
classes of the same names, as listed by importing the reference (tests/golden/api_surface.json, written by
tests/golden/make_api_surface.py).  Every public name of the reference class must exist here with the same kind (method /
property) and the reference's parameter names in the reference's order (extra trailing optional parameters are allowed); what is
deliberately absent is listed in WAIVED with its reason, so that an omission is a decision and not an accident."""
import inspect
import json
from pathlib import Path

import pytest

from moshi_amd import MimiModel, tiny_mimi_config
from moshi_amd.config import tiny_lm_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict

SURFACE = json.loads((Path(__file__).resolve().parent / "golden" / "api_surface.json").read_text())
OURS = {"MimiModel": MimiModel, "LMModel": LMModel, "LMGen": LMGen}

_HANDLE = "the engine's streaming state lives in the handle and is driven through LMGen (lm.py:646-665 does exactly that); "
WAIVED = {
    ("MimiModel", "__init__"): "built from a state dict + MimiConfig (the reference assembles nn.Modules); loaders.get_mimi keeps the reference's factory signature",
    ("LMModel", "__init__"): "built from a state dict + LMConfig; loaders.get_moshi_lm keeps the reference's factory signature",
    ("LMModel", "forward_text"): "internal to LMGen._step in the reference (lm.py:719); fused into mmi_lm_step",
    ("LMModel", "forward_depformer"): "internal to LMGen.depformer_step (lm.py:835); fused into mmi_lm_step",
    ("LMModel", "forward_depformer_training"): "training-mode forward: out of scope (DESIGN.md 9)",
    ("LMModel", "streaming"): _HANDLE + "no caller of the reference streams an LMModel directly",
    ("LMModel", "streaming_forever"): _HANDLE + "see streaming",
    ("LMModel", "reset_streaming"): _HANDLE + "see streaming",
    ("LMModel", "set_exec_mask"): _HANDLE + "see streaming",
    ("LMModel", "get_streaming_state"): _HANDLE + "LMGen.get_streaming_state returns the whole state",
    ("LMModel", "set_streaming_state"): _HANDLE + "see get_streaming_state",
    ("LMModel", "is_streaming"): _HANDLE + "LMGen.is_streaming",
    ("LMGen", "depformer_step"): "called only by LMGen._step in the reference (lm.py:760); the eight micro-steps are part of mmi_lm_step's launch list",
}


LOADERS_WAIVED = {
    "get_conditioner": "condition PROVIDERS (attributes -> tensors) are not on the frame step (DESIGN.md 9); their output enters as LMGen(condition_tensors=...)",
    "get_conditioner_provider": "builds a ConditionProvider out of get_conditioner results: see get_conditioner",
    "get_lora_moshi": "wraps nn.Linear modules in LoRA modules; the engine has no modules - get_moshi_lm(lora_weights=...) merges the adapter at load (loaders.py:486-516 with fuse_lora=True)",
}


@pytest.mark.parametrize("cls", sorted(OURS))
def test_public_surface_of_the_reference_class_exists_here(cls):
    ours = OURS[cls]
    missing, wrong = [], []
    for name, ref in SURFACE[cls].items():
        if (cls, name) in WAIVED:
            continue
        if not hasattr(ours, name) and name != "device":          # `device` is set per instance (checked below)
            missing.append(name)
            continue
        if ref["kind"] == "property":
            continue                                                # a property or an instance attribute: read below on an instance
        params = [p for p in inspect.signature(getattr(ours, name)).parameters if p != "self"]
        if ref.get("params") is not None and params[:len(ref["params"])] != ref["params"]:
            wrong.append((name, ref["params"], params))
    assert not missing, f"{cls}: public names of the reference missing here: {missing}"
    assert not wrong, f"{cls}: parameter names differ from the reference: {wrong}"


def test_waivers_name_things_the_reference_has():
    for (cls, name), reason in list(WAIVED.items()) + [(("module:loaders", n), r) for n, r in LOADERS_WAIVED.items()]:
        assert name in SURFACE[cls], f"waiver for {cls}.{name}, which the reference does not have"
        assert len(reason) > 20


def test_properties_and_caller_read_attributes_on_live_instances(sim_lib):
    """Every reference property reads on an instance, with the reference's values where they are constants of the format
    (lm.py:246-277: initial 2048-style ids = card / text_card, zero -1, ungenerated -2; compression.py:160-190), plus the instance
    attributes the reference's callers read (lm_model.{dep_q, delays, device, card, text_card, num_codebooks})."""
    mcfg, lcfg = tiny_mimi_config(), tiny_lm_config()
    mimi = MimiModel(random_mimi_state_dict(mcfg, seed=1), mcfg, device="cpu", max_batch=2, num_codebooks=4, lib=sim_lib)
    lm = LMModel(random_lm_state_dict(lcfg, seed=1), lcfg, device="cpu", max_batch=2, lib=sim_lib)
    gen = LMGen(lm)
    for cls, obj in (("MimiModel", mimi), ("LMModel", lm), ("LMGen", gen)):
        for name, ref in SURFACE[cls].items():
            if ref["kind"] == "property" and (cls, name) not in WAIVED:
                getattr(obj, name)
    assert (mimi.sample_rate, mimi.channels, mimi.cardinality, mimi.num_codebooks) == (mcfg.sample_rate, 1, mcfg.q_bins, 4)
    assert mimi.frame_size == int(mimi.sample_rate / mimi.frame_rate) and mimi.total_codebooks == mcfg.q_n_q
    assert (lm.zero_token_id, lm.ungenerated_token_id, lm.audio_offset) == (-1, -2, 1)
    assert (lm.initial_token_id, lm.text_initial_token_id) == (lcfg.card, lcfg.text_card)
    assert (lm.text_padding_token_id, lm.end_of_text_padding_id) == (3, 0)
    assert lm.num_codebooks == lcfg.n_q + 1 and lm.num_audio_codebooks == lcfg.n_q
    assert (lm.dep_q, lm.card, lm.text_card, list(lm.delays)) == (lcfg.dep_q, lcfg.card, lcfg.text_card, list(lcfg.delays))
    assert str(lm.device) == "cpu" and lm.condition_provider is None
    assert (gen.use_sampling, gen.temp, gen.temp_text) == (True, 0.8, 0.7)
    for obj in (mimi, lm, gen):
        obj.set_streaming_detached(True)
    assert not mimi.is_streaming and not gen.is_streaming



def test_loaders_module_surface():
    """`moshi.models.loaders` as its callers see it (server.py, run_inference.py, scripts): functions with the reference's
    parameter names first, the released file names as constants with the reference's values, CheckpointInfo's methods."""
    from moshi_amd import loaders
    for name, ref in SURFACE["module:loaders"].items():
        if name in LOADERS_WAIVED:
            continue
        assert hasattr(loaders, name), f"loaders.{name} is missing"
        ours = getattr(loaders, name)
        if ref["kind"] == "constant":
            assert ours == ref["value"], name
        elif ref["kind"] == "function":
            params = list(inspect.signature(ours).parameters)
            assert params[:len(ref["params"])] == ref["params"], (name, ref["params"], params)
        else:
            for m, rp in ref["methods"].items():
                params = [q for q in inspect.signature(getattr(ours, m)).parameters if q != "self"]
                rp = [q for q in rp if q != "kwargs"]
                assert params[:len(rp)] == rp, (name, m, rp, params)


def test_hf_get_resolves_local_names_and_refuses_downloads(tmp_path):
    from moshi_amd import loaders
    f = tmp_path / "model.safetensors"
    f.write_bytes(b"x")
    assert loaders.hf_get(f) == f                                                  # loaders.py:125-126
    assert loaders.hf_get(f"file://{f}") == f                                      # :132-135
    assert loaders.hf_get(str(f)) == f                                             # :141-142
    assert loaders.hf_get(str(f), loaders.DEFAULT_REPO, check_local_file_exists=True) == f     # :137-139
    for args in (("hf://kyutai/moshiko-pytorch-bf16/model.safetensors",), ("model.safetensors", loaders.DEFAULT_REPO)):
        with pytest.raises(RuntimeError, match="no network"):
            loaders.hf_get(*args)


def test_models_namespace_mirrors_the_reference_import_line():
    """server.py:24 / run_inference.py:20 of the reference: `from .models import loaders, MimiModel, LMModel, LMGen`."""
    from moshi_amd.models import LMGen as G, LMModel as L, MimiModel as M, get_mimi, get_moshi_lm, loaders  # noqa: F401
    assert (M, L, G) == (MimiModel, LMModel, LMGen) and loaders.get_mimi is get_mimi and loaders.get_moshi_lm is get_moshi_lm
