"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the build container only (it imports the read-only reference checkout; nothing at test time does):

    PYTHONPATH=/root/repo NO_TORCH_COMPILE=1 python tests/golden/make_golden.py [--ref /root/reference/moshi]

What is pinned and how
  mimi_tiny.npz   tiny Mimi (moshi_amd.config.tiny_mimi_config): the full synthetic state dict, 7 frames of
                  noise for 3 streams with a per-frame exec-mask schedule and a partial reset, and the reference's
                  latents / codes / PCM per frame (MimiModel.encode_to_latent / quantizer.encode / decode,
                  compression.py:338-433; exec-mask recipe of scripts/test_missing_data.py).
  mimi_full.npz   the real Mimi architecture (loaders._mimi_config) with synthetic weights that are NOT stored:
                  they are re-drawn from the seed by moshi_amd.weights (bit-reproducible for one torch build);
                  stored: input PCM, reference codes, latent and PCM for 4 frames x 2 streams.
  lm_tiny.npz     tiny Moshi LM: state dict, user codes, and LMGen.step outputs (greedy) + logits taps.
"""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent


def gen_mimi_tiny(loaders):
    from moshi_amd.config import tiny_mimi_config
    from moshi_amd.weights import random_mimi_state_dict
    cfg = tiny_mimi_config()
    sd = random_mimi_state_dict(cfg, seed=7)
    K = 4
    ref = loaders.get_mimi(None, cfg.reference_kwargs(), "cpu", num_codebooks=K)
    ref.load_state_dict(sd, strict=True)
    B, F = 3, 7
    g = torch.Generator().manual_seed(11)
    x = 0.4 * torch.randn(B, 1, cfg.frame_size * F, generator=g)
    # exec-mask schedule: row 1 skips frames 2 and 3, row 2 skips frame 4; rows {0,2} are reset before frame 5
    masks = np.ones((F, B), bool)
    masks[2, 1] = masks[3, 1] = False
    masks[4, 2] = False
    reset_before = {5: np.array([True, False, True])}
    lat, codes, pcm = [], [], []
    with torch.no_grad(), ref.streaming(B):
        for f in range(F):
            if f in reset_before:
                ref.reset_streaming(torch.from_numpy(reset_before[f]))
            ref.set_exec_mask(torch.from_numpy(masks[f]))
            xf = x[..., f * cfg.frame_size:(f + 1) * cfg.frame_size]
            l = ref.encode_to_latent(xf, quantize=False)
            c = ref.quantizer.encode(l)
            p = ref.decode(c)
            lat.append(l.numpy()); codes.append(c.numpy()); pcm.append(p.numpy())
    out = {"x": x.numpy(), "masks": masks, "reset_frame": np.array([5]), "reset_mask": reset_before[5],
           "latent": np.stack(lat), "codes": np.stack(codes), "pcm": np.stack(pcm), "num_codebooks": np.array([K])}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    np.savez_compressed(HERE / "mimi_tiny.npz", **out)
    print("mimi_tiny.npz", {k: v.shape for k, v in out.items() if not k.startswith("sd/")})


def gen_mimi_full(loaders):
    from moshi_amd.config import MimiConfig
    from moshi_amd.weights import random_mimi_state_dict
    cfg = MimiConfig()
    seed = 1234
    sd = random_mimi_state_dict(cfg, seed=seed)
    ref = loaders.get_mimi(None, None, "cpu", num_codebooks=8)
    ref.load_state_dict(sd, strict=True)
    B, F = 2, 4
    g = torch.Generator().manual_seed(5)
    t = torch.arange(cfg.frame_size * F) / cfg.sample_rate
    x = 0.2 * torch.randn(B, 1, cfg.frame_size * F, generator=g)
    x[1, 0] += 0.5 * torch.sin(2 * torch.pi * 440.0 * t)           # a sine row, as in SURVEY.md C1/C2
    lat, codes, pcm = [], [], []
    with torch.no_grad(), ref.streaming(B):
        for f in range(F):
            xf = x[..., f * cfg.frame_size:(f + 1) * cfg.frame_size]
            l = ref.encode_to_latent(xf, quantize=False)
            c = ref.quantizer.encode(l)
            p = ref.decode(c)
            lat.append(l.numpy()); codes.append(c.numpy()); pcm.append(p.numpy())
    # non-streaming encode of the same signal must give the same codes
    with torch.no_grad():
        c_batch = ref.encode(x).numpy()
    out = {"x": x.numpy(), "latent": np.stack(lat), "codes": np.stack(codes), "pcm": np.stack(pcm),
           "codes_nonstreaming": c_batch, "seed": np.array([seed])}
    np.savez_compressed(HERE / "mimi_full.npz", **out)
    print("mimi_full.npz", {k: v.shape for k, v in out.items()})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference/moshi")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.environ.setdefault("NO_TORCH_COMPILE", "1")
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    sys.path.insert(0, str(ROOT))
    from moshi.models import loaders  # the reference
    torch.set_num_threads(os.cpu_count() or 1)
    todo = [s for s in args.only.split(",") if s]
    if not todo or "mimi_tiny" in todo:
        gen_mimi_tiny(loaders)
    if not todo or "mimi_full" in todo:
        gen_mimi_full(loaders)
    if (not todo or "lm_tiny" in todo) and (HERE / "make_golden_lm.py").exists():
        from make_golden_lm import gen_lm_tiny  # type: ignore
        gen_lm_tiny()


if __name__ == "__main__":
    main()
