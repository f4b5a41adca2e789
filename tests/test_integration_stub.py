"""INTEGRATION.md section 2 promises a binding "a maintainer of the reference would add".  examples/reference_side_binding.py IS
that binding (ctypes + torch, nothing of moshi_amd imported); here it runs - on the simulator build of the library - from the
reference's own configuration dictionaries and state-dict keys, and must give, bit for bit, what moshi_amd's classes give."""
import importlib.util
import sys
from pathlib import Path

import numpy as np
import torch

from moshi_amd import MimiModel, tiny_mimi_config
from moshi_amd.config import tiny_lm_config
from moshi_amd.lm import LMGen, LMModel
from moshi_amd.weights import random_lm_state_dict, random_mimi_state_dict

ROOT = Path(__file__).resolve().parent.parent


def _binding():
    spec = importlib.util.spec_from_file_location("reference_side_binding", ROOT / "examples" / "reference_side_binding.py")
    mod = importlib.util.module_from_spec(spec)
    before = set(sys.modules)
    spec.loader.exec_module(mod)
    assert not [m for m in set(sys.modules) - before if m.startswith("moshi_amd")], "the reference-side binding must not import moshi_amd"
    return mod


def test_the_binding_of_integration_md_runs_and_equals_the_shipped_classes(sim_lib):
    b = _binding()
    lib = b.load(sim_lib.path)
    mcfg, lcfg = tiny_mimi_config(), tiny_lm_config()
    msd, lsd = random_mimi_state_dict(mcfg, seed=11), random_lm_state_dict(lcfg, seed=11)
    B, K = 3, lcfg.n_q - lcfg.dep_q
    mimi_a = b.MimiModelMI355X(msd, mcfg.reference_kwargs(), max_batch=B, num_codebooks=K, device="cpu", lib=lib)
    gen_a = b.LMGenMI355X(lsd, lcfg.reference_kwargs(), max_batch=B, use_sampling=False, device="cpu", lib=lib)
    mimi_b = MimiModel(msd, mcfg, device="cpu", max_batch=B, num_codebooks=K, lib=sim_lib)
    gen_b = LMGen(LMModel(lsd, lcfg, device="cpu", max_batch=B, lib=sim_lib), use_sampling=False)
    rng = np.random.default_rng(11)
    mask = torch.tensor([True, False, True])
    reset = torch.tensor([False, False, True])
    none_pattern = []
    with mimi_a.streaming(B), gen_a.streaming(B), mimi_b.streaming(B), gen_b.streaming(B):
        for f in range(6):
            if f == 3:
                for m in (mimi_a, gen_a, mimi_b, gen_b):
                    m.reset_streaming(reset)
            for m in (mimi_a, gen_a, mimi_b, gen_b):
                m.set_exec_mask(mask if f == 2 else torch.ones(B, dtype=torch.bool))
            pcm = torch.from_numpy((0.3 * rng.standard_normal((B, 1, mcfg.frame_size))).astype(np.float32))
            ca, cb = mimi_a.encode(pcm), mimi_b.encode(pcm)
            assert torch.equal(ca, cb), f"frame {f}: codes"
            # the tiny codec has 5 codebooks, the tiny LM listens to 8: the user stream is the codes with the first three repeated
            user = torch.cat([ca, ca[:, :K - ca.shape[1]]], 1) % lcfg.card
            ta, tb = gen_a.step(user), gen_b.step(user)
            none_pattern.append(ta is None)
            assert (ta is None) == (tb is None)
            if ta is not None:
                assert torch.equal(ta, tb), f"frame {f}: tokens"
                audio = ta[:, 1:1 + mimi_a.num_codebooks].clamp(0, mcfg.q_bins - 1)
                assert torch.equal(mimi_a.decode(audio).view(torch.int32), mimi_b.decode(audio).view(torch.int32)), f"frame {f}: pcm"
    # lm.py:774-776: None during the delay - at the start and again after the reset (lm.py:541-547 puts offset_cpu back to 0)
    assert none_pattern == [True, False, False, True, False, False]
    assert mimi_a.frame_size == mimi_b.frame_size and not mimi_a.is_streaming


def test_the_binding_maps_statuses_to_the_reference_exception_types(sim_lib):
    import pytest
    b = _binding()
    lib = b.load(sim_lib.path)
    mcfg, lcfg = tiny_mimi_config(), tiny_lm_config()
    gen = b.LMGenMI355X(random_lm_state_dict(lcfg, seed=1), lcfg.reference_kwargs(), max_batch=2, device="cpu", lib=lib)
    with pytest.raises(RuntimeError, match="streaming"):                      # lm.py:673-676
        gen.step(torch.zeros(2, 8, 1, dtype=torch.long))
    with gen.streaming(2):
        with pytest.raises(AssertionError):                                   # batch mismatch (lm.py:681)
            gen.step(torch.zeros(1, 8, 1, dtype=torch.long))
    sd = random_mimi_state_dict(mcfg, seed=1)
    sd.pop(next(iter(sd)))
    with pytest.raises(KeyError):                                             # a missing state-dict key
        b.MimiModelMI355X(sd, mcfg.reference_kwargs(), max_batch=2, device="cpu", lib=lib)
