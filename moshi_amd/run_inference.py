"""Offline batched inference: the reference's `python -m moshi.run_inference` on the engine
(moshi/moshi/run_inference.py:60-217 `InferenceState`, :220-330 `main`).

    python -m moshi_amd.run_inference --moshi-weight model.safetensors --mimi-weight mimi.safetensors \
        --tokenizer tokenizer_spm_32k_3.model [--config config.json] [--batch-size 8] in.wav out.wav

Same loop as the reference: the input is cut into whole 80 ms frames, every frame goes `mimi.encode -> lm_gen.step ->
mimi.decode`, the first frame is stepped twice so that the transformer sees it (run_inference.py:160-166), the Hibiki
translation model gets an end-of-stream code (2048 on every codebook) followed by silence until every item has emitted
the text EOS (run_inference.py:137-155, 178-186).  Audio files are read and written with scipy (16-bit or float WAV at
the codec's sample rate); there is no resampler and no network path: all files are local.
"""
from __future__ import annotations

import argparse
import json
import sys
import time
from collections import deque
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import loaders
from .lm import LMGen, LMModel
from .mimi import MimiModel


def log(level: str, msg: str) -> None:
    print(f"[{level}] {msg}", file=sys.stderr, flush=True)


class InferenceState:
    """run_inference.py:60-94.  `condition_tensors` replaces the reference's `get_condition_tensors(model_type, lm, ...)`:
    condition providers run outside the engine, their output tensors are passed in."""

    def __init__(self, checkpoint_info, mimi: MimiModel, text_tokenizer, lm: LMModel, batch_size: int, cfg_coef: float = 1.0,
                 device: str | torch.device = "cuda", condition_tensors=None, on_token=None, **kwargs):
        self.checkpoint_info = checkpoint_info
        self.model_type = getattr(checkpoint_info, "model_type", "moshi")
        self.mimi = mimi
        self.text_tokenizer = text_tokenizer
        self.lm_gen = LMGen(lm, cfg_coef=cfg_coef, condition_tensors=condition_tensors, **kwargs)
        self.device = torch.device(device)
        self.frame_size = int(self.mimi.sample_rate / self.mimi.frame_rate)
        self.batch_size = batch_size
        self.on_token = on_token or (lambda text: print(text, end="", flush=True))
        self.mimi.streaming_forever(batch_size)
        self.lm_gen.streaming_forever(batch_size)

    def run(self, in_pcms: torch.Tensor, max_steps: Optional[int] = None) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """in_pcms float [B, channels, T] -> per item (text tokens [n], pcm [channels, n * frame_size]).  `max_steps` bounds the
        Hibiki wait-for-EOS loop (not in the reference, which trusts the model to emit EOS)."""
        B = self.batch_size
        out_pcms: List[List[torch.Tensor]] = [[] for _ in range(B)]
        out_text: List[List[torch.Tensor]] = [[] for _ in range(B)]
        eos_reached = [False] * B
        need_eos_input = True
        lm = self.lm_gen.lm_model
        log("info", f"starting inference, sampling: {self.lm_gen.use_sampling}, audio temp: {self.lm_gen.temp}, "
                    f"text temp: {self.lm_gen.temp_text}")
        start, ntokens, first_frame = time.time(), 0, True
        chunks = deque(c for c in in_pcms.split(self.frame_size, dim=2) if c.shape[-1] == self.frame_size)   # whole frames only
        eos_id = self.text_tokenizer.eos_id() if self.text_tokenizer is not None else -1
        while not all(eos_reached):
            if chunks:
                codes = self.mimi.encode(chunks.popleft().to(self.device))
            elif self.model_type == "hibiki":
                if need_eos_input:           # first frame after the end of the file: a code full of `cardinality` = end of stream
                    need_eos_input = False
                    codes = torch.full((B, self.mimi.num_codebooks, 1), self.mimi.cardinality, device=self.device, dtype=torch.long)
                else:
                    codes = self.mimi.encode(torch.zeros(B, self.mimi.channels, self.frame_size, device=self.device))
            else:
                break                        # other models stop at the end of the audio
            if first_frame:                  # run_inference.py:160-166
                tokens = self.lm_gen.step(codes)
                if max(lm.delays) > 0:
                    assert tokens is None
                first_frame = False
            tokens = self.lm_gen.step(codes)
            if tokens is None:
                continue
            assert tokens.shape[1] == lm.dep_q + 1
            out_pcm = self.mimi.decode(tokens[:, 1:]).cpu()
            for b, (one_text, one_pcm) in enumerate(zip(tokens[:, 0].cpu(), out_pcm)):
                if eos_reached[b]:
                    continue
                if one_text.item() == eos_id:
                    if need_eos_input:
                        log("warning", "EOS sampled too early.")
                    else:
                        eos_reached[b] = True
                out_text[b].append(one_text)
                out_pcms[b].append(one_pcm)
                if b == 0 and one_text.item() not in (0, 3) and self.text_tokenizer is not None:
                    self.on_token(self.text_tokenizer.id_to_piece(one_text.item()).replace("▁", " "))
            ntokens += 1
            if max_steps is not None and ntokens >= max_steps:
                break
        dt = time.time() - start
        if ntokens:
            log("info", f"processed {ntokens} steps in {dt:.0f}s, {1000 * dt / ntokens:.2f}ms/step")
        return [(torch.cat(t, dim=0) if t else torch.zeros(0, dtype=torch.long),
                 torch.cat(p, dim=1) if p else torch.zeros(self.mimi.channels, 0)) for t, p in zip(out_text, out_pcms)]


def read_wav(path: str, sample_rate: int) -> np.ndarray:
    from scipy.io import wavfile
    sr, data = wavfile.read(path)
    if sr != sample_rate:
        raise ValueError(f"{path} is sampled at {sr} Hz; the codec runs at {sample_rate} Hz and the engine has no resampler")
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    data = data.astype(np.float32)
    return data.mean(axis=1) if data.ndim == 2 else data     # mono, like sphn.read(...)[0].mean(axis=0)


def write_wav(path: str, pcm: np.ndarray, sample_rate: int) -> None:
    from scipy.io import wavfile
    wavfile.write(path, sample_rate, (np.clip(pcm, -1, 1) * 32767.0).astype(np.int16))


def main(argv: Optional[List[str]] = None) -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokenizer", type=str, help="Path to a local tokenizer file.")
    ap.add_argument("--moshi-weight", type=str, help="Path to a local checkpoint file for Moshi.")
    ap.add_argument("--mimi-weight", type=str, help="Path to a local checkpoint file for Mimi.")
    ap.add_argument("--checkpoint-dir", type=str, help="A downloaded model repository (config.json + the files it names).")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--config", "--lm-config", dest="config", type=str, help="The config as a json file.")
    ap.add_argument("--cfg-coef", type=float, default=1.0)
    ap.add_argument("--quantize", choices=["none", "int8", "fp8"], default="none")
    ap.add_argument("--greedy", action="store_true")
    ap.add_argument("infile", type=str)
    ap.add_argument("outfile", type=str)
    args = ap.parse_args(argv)

    if args.checkpoint_dir:
        info = loaders.CheckpointInfo.from_local(args.checkpoint_dir, args.moshi_weight, args.mimi_weight, args.tokenizer)
    else:
        lm_config = json.loads(open(args.config).read()) if args.config else None
        from pathlib import Path
        info = loaders.CheckpointInfo(Path(args.moshi_weight), Path(args.mimi_weight), Path(args.tokenizer) if args.tokenizer else None,
                                      lm_config=lm_config)
    log("info", "loading mimi")
    mimi = info.get_mimi(device=args.device, max_batch=args.batch_size)
    log("info", "loading moshi")
    lm = info.get_moshi(device=args.device, max_batch=args.batch_size * (2 if args.cfg_coef != 1.0 else 1),
                        quantize=None if args.quantize == "none" else args.quantize)
    tokenizer = None
    if info.tokenizer is not None:
        import sentencepiece
        tokenizer = sentencepiece.SentencePieceProcessor(str(info.tokenizer))
    state = InferenceState(info, mimi, tokenizer, lm, args.batch_size, args.cfg_coef, args.device,
                           use_sampling=not args.greedy, **info.lm_gen_config)
    pcm = torch.from_numpy(read_wav(args.infile, mimi.sample_rate))[None, None]
    out = state.run(pcm.expand(args.batch_size, -1, -1).contiguous())
    from pathlib import Path
    stem = Path(args.outfile)
    for b, (_, one_pcm) in enumerate(out):
        name = stem if args.batch_size == 1 else stem.with_name(f"{stem.stem}-{b}{stem.suffix}")
        write_wav(str(name), one_pcm[0].numpy(), mimi.sample_rate)
        log("info", f"wrote {name}")


if __name__ == "__main__":
    with torch.no_grad():
        main()
