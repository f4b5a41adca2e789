"""Parity cases shared by the simulator tests (CPU) and the GPU tests: engine vs golden vectors / oracle."""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from moshi_amd import MimiConfig, MimiModel, tiny_mimi_config
from moshi_amd.weights import random_mimi_state_dict
from oracle.mimi_oracle import MimiOracle

GOLDEN = Path(__file__).resolve().parent / "golden"

# PCM tolerance: fp32 everywhere, only the summation order differs from the reference (MFMA k-order vs MKL/oneDNN).
# Stated as max |diff| <= PCM_ATOL + PCM_RTOL * max|ref| per frame.
PCM_ATOL, PCM_RTOL = 2e-5, 2e-5
LATENT_ATOL, LATENT_RTOL = 2e-5, 2e-5


def close(a: np.ndarray, b: np.ndarray, atol: float, rtol: float) -> bool:
    return float(np.abs(a - b).max()) <= atol + rtol * float(np.abs(b).max())


def load_tiny():
    g = np.load(GOLDEN / "mimi_tiny.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    return g, sd


def run_tiny_schedule(model_factory, device):
    """Replays the golden schedule (exec masks + partial reset) through an engine; returns per-frame outputs."""
    g, sd = load_tiny()
    cfg = tiny_mimi_config()
    K = int(g["num_codebooks"][0])
    m = model_factory(sd, cfg, K)
    x = torch.from_numpy(g["x"]).to(device)
    F, B = g["masks"].shape
    fs = cfg.frame_size
    lat, codes, pcm = [], [], []
    with m.streaming(B):
        for f in range(F):
            if f == int(g["reset_frame"][0]):
                m.reset_streaming(torch.from_numpy(g["reset_mask"]).to(device))
            m.set_exec_mask(torch.from_numpy(g["masks"][f]).to(device))
            xf = x[..., f * fs:(f + 1) * fs]
            c = m.encode(xf)
            # decode the REFERENCE codes so that a (legitimate) near-tie flip cannot cascade into the PCM check
            p = m.decode(torch.from_numpy(g["codes"][f]).to(device))
            codes.append(c.cpu().numpy()); pcm.append(p.cpu().numpy())
    return g, np.stack(codes), np.stack(pcm)


def check_tiny_against_golden(model_factory, device):
    g, codes, pcm = run_tiny_schedule(model_factory, device)
    masks = g["masks"]
    F, B = masks.shape
    for f in range(F):
        for b in range(B):
            if not masks[f, b]:
                continue        # rows that did not execute produce don't-care output (streaming.py:183-195)
            assert np.array_equal(codes[f, b], g["codes"][f, b]), f"codes differ at frame {f} row {b}"
            assert close(pcm[f, b], g["pcm"][f, b], PCM_ATOL, PCM_RTOL), f"pcm differs at frame {f} row {b}"


def check_full_against_golden(model_factory, device, frames=None):
    g = np.load(GOLDEN / "mimi_full.npz")
    cfg = MimiConfig()
    sd = random_mimi_state_dict(cfg, seed=int(g["seed"][0]))
    m = model_factory(sd, cfg, 8)
    x = torch.from_numpy(g["x"]).to(device)
    F = g["codes"].shape[0] if frames is None else frames
    B = x.shape[0]
    fs = cfg.frame_size
    with m.streaming(B):
        for f in range(F):
            xf = x[..., f * fs:(f + 1) * fs]
            lat = m.encode_to_latent(xf, quantize=False).cpu().numpy()
            assert close(lat, g["latent"][f], LATENT_ATOL, LATENT_RTOL), f"latent differs at frame {f}"
            c = m.quantize(torch.from_numpy(g["latent"][f]).to(device)).cpu().numpy()
            assert np.array_equal(c, g["codes"][f]), f"RVQ indices differ on the reference latent at frame {f}"
            p = m.decode(torch.from_numpy(g["codes"][f]).to(device)).cpu().numpy()
            assert close(p, g["pcm"][f], PCM_ATOL, PCM_RTOL), f"pcm differs at frame {f}"
    return m


def oracle_vs_engine(model_factory, device, cfg, seed, B, F, K, use_masks=True):
    """Seeded comparison of an engine with the numpy oracle on identical inputs."""
    sd = random_mimi_state_dict(cfg, seed=seed)
    m = model_factory(sd, cfg, K)
    orc = MimiOracle(sd, cfg, num_codebooks=K)
    rng = np.random.default_rng(seed)
    x = (0.3 * rng.standard_normal((B, 1, cfg.frame_size * F))).astype(np.float32)
    orc.streaming(B)
    fs = cfg.frame_size
    with m.streaming(B):
        for f in range(F):
            mask = np.ones(B, bool)
            if use_masks and B > 1:
                mask = rng.random(B) > 0.3
                mask[0] = True
            if use_masks and f == F // 2 and B > 1:
                rmask = np.zeros(B, bool); rmask[B - 1] = True
                orc.reset_streaming(rmask)
                m.reset_streaming(torch.from_numpy(rmask).to(device))
                mask[B - 1] = True
            orc.set_exec_mask(mask)
            m.set_exec_mask(torch.from_numpy(mask).to(device))
            xf = x[..., f * fs:(f + 1) * fs]
            co = orc.encode(xf)
            ce = m.encode(torch.from_numpy(xf).to(device)).cpu().numpy()
            po = orc.decode(co)
            pe = m.decode(torch.from_numpy(co).to(device)).cpu().numpy()
            for b in range(B):
                if not mask[b]:
                    continue
                assert np.array_equal(ce[b], co[b]), f"codes differ frame {f} row {b}: {ce[b].ravel()} vs {co[b].ravel()}"
                assert close(pe[b], po[b], PCM_ATOL, PCM_RTOL), f"pcm differs frame {f} row {b}"
