"""The numpy oracle is pinned against golden vectors produced by running the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from moshi_amd import MimiConfig, tiny_mimi_config
from moshi_amd.weights import random_mimi_state_dict
from oracle.mimi_oracle import MimiOracle
from tests.mimi_cases import GOLDEN, close, load_tiny


def test_mimi_oracle_tiny_schedule_matches_reference():
    g, sd = load_tiny()
    cfg = tiny_mimi_config()
    K = int(g["num_codebooks"][0])
    orc = MimiOracle(sd, cfg, num_codebooks=K)
    F, B = g["masks"].shape
    orc.streaming(B)
    fs = cfg.frame_size
    for f in range(F):
        if f == int(g["reset_frame"][0]):
            orc.reset_streaming(g["reset_mask"])
        orc.set_exec_mask(g["masks"][f])
        xf = g["x"][..., f * fs:(f + 1) * fs]
        lat = orc.encode_to_latent(xf)
        codes = orc.quantize(lat)
        pcm = orc.decode(g["codes"][f])
        for b in range(B):
            if not g["masks"][f, b]:
                continue
            assert close(lat[b], g["latent"][f, b], 1e-5, 1e-5), (f, b)
            assert np.array_equal(codes[b], g["codes"][f, b]), (f, b)
            assert close(pcm[b], g["pcm"][f, b], 1e-5, 1e-5), (f, b)


def test_mimi_oracle_full_size_matches_reference():
    g = np.load(GOLDEN / "mimi_full.npz")
    cfg = MimiConfig()
    sd = random_mimi_state_dict(cfg, seed=int(g["seed"][0]))
    orc = MimiOracle(sd, cfg, num_codebooks=8)
    B = g["x"].shape[0]
    orc.streaming(B)
    fs = cfg.frame_size
    for f in range(2):
        xf = g["x"][..., f * fs:(f + 1) * fs]
        lat = orc.encode_to_latent(xf)
        assert close(lat, g["latent"][f], 2e-5, 2e-5)
        assert np.array_equal(orc.quantize(g["latent"][f]), g["codes"][f])
        assert close(orc.decode(g["codes"][f]), g["pcm"][f], 2e-5, 2e-5)
    # streaming codes == non-streaming codes in the reference
    assert np.array_equal(np.concatenate(list(g["codes"]), -1), g["codes_nonstreaming"])
