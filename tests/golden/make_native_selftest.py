"""Writes tests/golden/native_selftest/{manifest.txt,expected.bin}: what scripts/native_selftest.cpp needs to check an installed
libmoshi_mi.so on a GPU box WITHOUT Python - a tiny Mimi and a tiny Moshi LM whose weights and inputs both sides generate from the
same integer hash (so no weight file travels), and the outputs the numpy ORACLE (oracle/, test infrastructure) computes for them.

    python tests/golden/make_native_selftest.py

manifest.txt is line based:
    mimi_cfg / lm_cfg <fields of mmi_mimi_cfg / mmi_lm_cfg in header order>
    T <model> <name> <f32|bf16> <ndim> <d0> <d1> <d2> <d3> <base> <scale> <seed>      value[i] = base + scale * u(seed, i)
    run <B> <frames> <K> <steps> ...
    E <name> <i64|f32> <count> <byte offset into expected.bin>
u(seed, i) in [-1, 1): h = (i * 2654435761) ^ seed; h ^= h >> 13; h *= 0x5bd1e995; h ^= h >> 15 (uint32);
u = float((h & 0xffff) - 32768) * 2^-15; every step in fp32, bf16 by round-to-nearest-even."""
import math
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
from moshi_amd import tiny_mimi_config  # noqa: E402
from moshi_amd.config import tiny_lm_config  # noqa: E402
from moshi_amd.weights import lm_state_spec, mimi_state_spec  # noqa: E402
from oracle.lm_oracle import LMOracle  # noqa: E402
from oracle.mimi_oracle import MimiOracle  # noqa: E402

OUT = Path(__file__).resolve().parent / "native_selftest"
B, FRAMES, STEPS = 2, 3, 5


def u(seed: int, n: int) -> np.ndarray:
    i = np.arange(n, dtype=np.uint64)
    h = ((i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)).astype(np.uint32) ^ np.uint32(seed)
    h ^= h >> np.uint32(13)
    h = ((h.astype(np.uint64) * np.uint64(0x5BD1E995)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    return ((h & np.uint32(0xFFFF)).astype(np.int32) - 32768).astype(np.float32) * np.float32(2.0 ** -15)


def rule(kind: str):
    """(base, scale) of a tensor kind of moshi_amd/weights.py's specs (uniform stand-ins for its normal draws)."""
    if kind.startswith("fan:") or kind.startswith("emb:"):
        return 0.0, math.sqrt(3.0 / int(kind[4:]))
    return {"codebook": (0.0, 1.7), "usage": (1.25, 0.75), "norm_w": (1.0, 0.17), "alpha": (1.0, 0.17), "norm_b": (0.0, 0.087),
            "layer_scale": (0.3, 0.087), "one": (1.0, 0.0)}[kind if not kind.startswith("codebook") else "codebook"]


def build(spec, dtype, model, seed0, lines):
    sd = {}
    for j, (name, shape, kind) in enumerate(spec):
        base, scale = rule(kind)
        seed = seed0 + 7919 * j
        n = int(np.prod(shape))
        v = np.float32(base) + np.float32(scale) * u(seed, n)
        t = torch.from_numpy(v.reshape(shape).copy())
        sd[name] = t.to(dtype)
        dims = list(shape) + [1] * (4 - len(shape))
        lines.append(f"T {model} {name} {'bf16' if dtype == torch.bfloat16 else 'f32'} {len(shape)} {dims[0]} {dims[1]} {dims[2]} {dims[3]} "
                     f"{float(np.float32(base)).hex()} {float(np.float32(scale)).hex()} {seed}")
    return sd


def main():
    OUT.mkdir(exist_ok=True)
    mcfg, lcfg = tiny_mimi_config(), tiny_lm_config()
    K = 5
    lines = []
    from moshi_amd.mimi import _mimi_cfg_struct
    from moshi_amd.lm import _lm_cfg_struct
    ms, ls = _mimi_cfg_struct(mcfg), _lm_cfg_struct(lcfg)

    def fields(s):
        out = []
        for name, ctype in s._fields_:
            v = getattr(s, name)
            out += [str(x) for x in v] if hasattr(v, "__len__") else [repr(float(v)) if isinstance(v, float) else str(v)]
        return " ".join(out)
    lines.append("mimi_cfg " + fields(ms))
    lines.append("lm_cfg " + fields(ls))
    msd = build(mimi_state_spec(mcfg), torch.float32, "mimi", 1000, lines)
    lsd = build(lm_state_spec(lcfg), torch.bfloat16, "lm", 500000, lines)
    fs = mcfg.frame_size
    n_user = lcfg.n_q - lcfg.dep_q
    lines.append(f"run {B} {FRAMES} {K} {STEPS} {fs} {n_user} {lcfg.dep_q} {lcfg.card} {lcfg.text_card}")

    # ---- Mimi: FRAMES frames of 0.3 * u(77 + f, B * fs); expected codes, and the PCM decoded from those codes
    orc = MimiOracle(msd, mcfg, num_codebooks=K)
    orc.streaming(B)
    codes, pcm = [], []
    for f in range(FRAMES):
        x = (np.float32(0.3) * u(77 + f, B * fs)).reshape(B, 1, fs)
        c = orc.encode(x)
        codes.append(c.reshape(B, K))
        pcm.append(orc.decode(c).reshape(B, fs))
    # ---- LM: greedy, teacher-forced with the oracle's own tokens; user codes = hash mod card
    lo = LMOracle(lsd, lcfg)
    lo.streaming(B)
    forced, outs, tls, als = [], [], [], []
    for s in range(STEPS):
        uc = (((u(9000 + s, B * n_user) + 1.0) * 32768.0).astype(np.int64) % lcfg.card).reshape(B, n_user, 1)
        oo, (tl, al, tt, at) = lo.step(uc, use_sampling=False, support_out_of_sync=True)
        forced.append(np.concatenate([tt[:, None], at], 1).astype(np.int64))
        outs.append(oo.reshape(B, lcfg.dep_q + 1).astype(np.int64))
        tls.append(tl.astype(np.float32))
        als.append(al.astype(np.float32))
    blobs = [("mimi_codes", np.stack(codes).astype(np.int64)), ("mimi_pcm", np.stack(pcm).astype(np.float32)),
             ("lm_forced", np.stack(forced)), ("lm_out", np.stack(outs)), ("lm_text_logits", np.stack(tls)), ("lm_audio_logits", np.stack(als))]
    off, data = 0, b""
    for name, a in blobs:
        raw = np.ascontiguousarray(a).tobytes()
        lines.append(f"E {name} {'i64' if a.dtype == np.int64 else 'f32'} {a.size} {off}")
        data += raw
        off += len(raw)
    (OUT / "manifest.txt").write_text("\n".join(lines) + "\n")
    (OUT / "expected.bin").write_bytes(data)
    print(OUT, len(lines), "lines,", len(data), "bytes; mimi codes", np.stack(codes)[0].ravel()[:6], "lm out", outs[-1][0])


if __name__ == "__main__":
    # python tests/golden/make_native_selftest.py [batch [out_dir]]   (default: 2 sessions -> native_selftest/; 18 sessions, the 32-row
    # batch tile and the wide-batch conv kernels -> native_selftest_b18/)
    if len(sys.argv) > 1:
        B = int(sys.argv[1])
        OUT = Path(sys.argv[2]) if len(sys.argv) > 2 else OUT.parent / f"native_selftest_b{B}"
    main()
