"""Does the serial frame loop run at the same speed on a non-default torch stream?  (diagnostics for the duplex pipeline)"""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


class A:
    lm_layers = 0; quant = "none"; kv = "bf16"


def main():
    dev = torch.device("cuda", 0)
    from bench_lm import make_lm
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.weights import random_mimi_state_dict
    B = 32
    mcfg = MimiConfig()
    mimi = MimiModel(random_mimi_state_dict(mcfg, seed=1234, device=dev), mcfg, device=dev, max_batch=B, num_codebooks=8)
    mimi.streaming_forever(B)
    gen = make_lm(dev, B, A)
    pcm = 0.1 * torch.randn(B, 1, mcfg.frame_size, device=dev)
    codes0 = torch.randint(0, 2048, (B, 8, 1), device=dev)

    def frame():
        codes = mimi.encode(pcm); tok = gen.step(codes)
        return None if tok is None else mimi.decode(tok[:, 1:])

    def lm_only():
        return gen.step(codes0)

    def mimi_only():
        return mimi.decode(mimi.encode(pcm))
    streams = [("default stream", None), ("torch.cuda.Stream()", torch.cuda.Stream(dev)), ("torch.cuda.Stream(priority=-1)", torch.cuda.Stream(dev, priority=-1)),
               ("default stream again", None)]
    for name, st in streams:
        for what, fn in (("frame", frame), ("lm", lm_only), ("mimi", mimi_only)):
            ctx = torch.cuda.stream(st) if st is not None else torch.cuda.stream(torch.cuda.default_stream(dev))
            with ctx:
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 30
            print(f"{name:32s} {what:6s} {1e3*dt:.3f} ms per call (30 back to back)", flush=True)


if __name__ == "__main__":
    main()
