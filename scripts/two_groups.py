"""Experiment: G independent session groups per GPU, each with its own Mimi + Moshi-7B handles and duplex pipeline, stepped from G
host threads: does the depth-transformer / codec phase of one group hide under the HBM-bound temporal phase of another?"""
import argparse
import sys
import threading
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--groups", type=int, default=2)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--lm-layers", type=int, default=0)
    ap.add_argument("--quant", default="none")
    ap.add_argument("--kv", default="bf16")
    ap.add_argument("--serial", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from bench_lm import make_lm
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.weights import random_mimi_state_dict
    B = args.batch
    mcfg = MimiConfig()
    groups = []
    for g in range(args.groups):
        mimi = MimiModel(random_mimi_state_dict(mcfg, seed=1234, device=dev), mcfg, device=dev, max_batch=B, num_codebooks=8)
        mimi.streaming_forever(B)
        gen = make_lm(dev, B, args)
        groups.append((mimi, gen, None if args.serial else DuplexStream(mimi, gen), torch.cuda.Stream(dev)))
    pcm = 0.1 * torch.randn(B, 1, mcfg.frame_size, device=dev)
    torch.cuda.synchronize()

    def loop(g, n):
        mimi, gen, dup, st = groups[g]
        with torch.cuda.stream(st):
            for _ in range(n):
                if dup is not None:
                    dup.step(pcm, want_tokens=False)
                else:
                    tok = gen.step(mimi.encode(pcm))
                    if tok is not None:
                        mimi.decode(tok[:, 1:])
            if dup is not None:
                dup.flush()
            st.synchronize()

    def run(n):
        th = [threading.Thread(target=loop, args=(g, n)) for g in range(args.groups)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(40)
    dt = run(args.steps)
    fr = args.groups * B * args.steps / dt
    print(f"groups {args.groups} x {B} sessions, {'serial' if args.serial else 'pipelined'}, quant {args.quant}: "
          f"{1e3*dt/args.steps:.3f} ms per round of {args.groups*B} frames, {fr:.0f} frames/s", flush=True)


if __name__ == "__main__":
    main()
