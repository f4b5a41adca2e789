"""Where does a frame's time go in the duplex pipeline?  Per-phase device timestamps (mmi_duplex_get_timeline) of isolated frames
and of the last frame of a back-to-back run, on the benchmark models (Mimi + Moshi-7B, 32 sessions).  GPU only; diagnostics."""
import argparse
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--lm-layers", type=int, default=0)
    ap.add_argument("--quant", default="none")
    ap.add_argument("--kv", default="bf16")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from bench_lm import make_lm
    from moshi_amd import MimiConfig, MimiModel
    from moshi_amd.duplex import DuplexStream
    from moshi_amd.weights import random_mimi_state_dict
    B = args.batch
    mcfg = MimiConfig()
    mimi = MimiModel(random_mimi_state_dict(mcfg, seed=1234, device=dev), mcfg, device=dev, max_batch=B, num_codebooks=8)
    mimi.streaming_forever(B)
    gen = make_lm(dev, B, args)
    dup = DuplexStream(mimi, gen)
    pcm = 0.1 * torch.randn(B, 1, mcfg.frame_size, device=dev)
    for _ in range(6):
        dup.step(pcm, want_tokens=False)
    dup.join(); torch.cuda.synchronize()
    dup.timeline(True)

    def show(tag, t):
        e, l, d = t["encode"], t["lm"], t["decode"]
        print(f"{tag}: encode {e[0]:.3f}->{e[1]:.3f} ({e[1]-e[0]:.3f})  lm {l[0]:.3f}->{l[1]:.3f} ({l[1]-l[0]:.3f})  "
              f"decode {d[0]:.3f}->{d[1]:.3f} ({d[1]-d[0]:.3f})", flush=True)
    for i in range(4):
        dup.step(pcm, want_tokens=False)
        dup.join(); torch.cuda.synchronize()
        show(f"isolated frame {i}", dup.timeline())
    for i in range(3):
        dup.step(pcm, want_tokens=False)
        time.sleep(0.03)                      # the frame is long done: no decode / caller wave polls while it runs
        dup.join(); torch.cuda.synchronize()
        show(f"isolated frame, join 30 ms later {i}", dup.timeline())
    # serial loop for comparison, same process
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3):
        a.record()
        codes = mimi.encode(pcm); tok = gen.step(codes); out = mimi.decode(tok[:, 1:])
        b.record(); torch.cuda.synchronize()
        print(f"serial frame {i}: {a.elapsed_time(b):.3f} ms", flush=True)
    # host cost of the calls
    torch.cuda.synchronize()
    for name, fn in (("mimi.encode", lambda: mimi.encode(pcm)), ("lm_gen.step", lambda: gen.step(mimi_codes)), ("mimi.decode", lambda: mimi.decode(toks[:, 1:]))):
        mimi_codes = mimi.encode(pcm); toks = gen.step(mimi_codes); torch.cuda.synchronize()
        hs = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); hs.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        print(f"host time of {name} (device idle): median {1e3*sorted(hs)[5]:.3f} ms", flush=True)
    hs = []
    for _ in range(20):
        t0 = time.perf_counter(); dup.step(pcm, want_tokens=False); hs.append(time.perf_counter() - t0)
    dup.join(); torch.cuda.synchronize()
    print("host time of dup.step back to back (includes flow-control waits): " + " ".join(f"{1e3*h:.2f}" for h in hs), flush=True)
    hs = []
    for _ in range(6):
        dup.join(); torch.cuda.synchronize()
        t0 = time.perf_counter(); dup.step(pcm, want_tokens=False); hs.append(time.perf_counter() - t0)
    dup.join(); torch.cuda.synchronize()
    print("host time of dup.step (device idle): " + " ".join(f"{1e3*h:.2f}" for h in hs), flush=True)
    for n in (8, 9, 20):
        t0 = time.perf_counter()
        for _ in range(n):
            dup.step(pcm, want_tokens=False)
        dup.join(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        show(f"last of {n} back-to-back frames ({1e3*dt/n:.3f} ms/frame wall)", dup.timeline())
    # device-clock stamps of the last four frames of that run on one time base: where the LM's queue waits between two steps
    st = dup.stamps()
    names = DuplexStream.STAMPS
    print("device-clock stamps (ms), last four frames of the 20-frame run:")
    print("frame " + " ".join(f"{n:>8s}" for n in names))
    for f in sorted(st):
        print(f"{f:5d} " + " ".join(f"{st[f][n]:8.3f}" if n in st[f] else "       -" for n in names))
    fr = sorted(st)
    for a_, b_ in zip(fr, fr[1:]):
        x, y = st[a_], st[b_]
        if "lm1" in x and "lm0" in y:
            print(f"frame {b_}: LM idle since step {a_} ended {y['lm0'] - x['lm1']:.3f} ms; its wait for encode({b_}) took "
                  f"{y.get('wait1', 0) - y.get('wait0', 0):.3f} ms (began {y.get('wait0', 0) - x['lm1']:.3f} ms after that end); "
                  f"encode({b_}) ended {y.get('enc1', 0) - x['lm1']:+.3f} ms relative to it; phase({a_}) -> end {x['lm1'] - x.get('phase', 0):.3f} ms; "
                  f"LM begin -> phase {y.get('phase', 0) - y['lm0']:.3f} ms")


if __name__ == "__main__":
    main()
