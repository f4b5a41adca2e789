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

    def _user_codes(self, in_pcms: torch.Tensor, finished):
        """The user-side code stream: one [B, K, 1] tensor per whole input frame; for Hibiki, after the audio, one
        end-of-stream frame (every codebook = cardinality) and then encoded silence for as long as `finished()` is false
        (run_inference.py:137-155).  The second element tells whether the input audio is exhausted."""
        F, B = self.frame_size, self.batch_size
        n_whole = in_pcms.shape[-1] // F                       # a ragged tail is dropped (run_inference.py:128-134)
        for f in range(n_whole):
            yield self.mimi.encode(in_pcms[..., f * F:(f + 1) * F].to(self.device)), False
        if self.model_type != "hibiki":
            return
        yield torch.full((B, self.mimi.num_codebooks, 1), self.mimi.cardinality, device=self.device, dtype=torch.long), True
        silence = torch.zeros(B, self.mimi.channels, F, device=self.device)
        while not finished():
            yield self.mimi.encode(silence), True

    def run(self, in_pcms: torch.Tensor, max_steps: Optional[int] = None) -> List[Tuple[torch.Tensor, torch.Tensor]]:
        """in_pcms float [B, channels, T] -> per item (text tokens [n], pcm [channels, n * frame_size]).  `max_steps` bounds the
        Hibiki wait-for-EOS loop (not in the reference, which trusts the model to emit EOS)."""
        B, lm = self.batch_size, self.lm_gen.lm_model
        if self.model_type == "stt":
            # the text stream runs `audio_delay_seconds` behind the audio: pad the input so that the last words still come out
            # (run_inference.py:121-127: left = audio_silence_prefix_seconds, right = audio_delay_seconds + 1 s, at 24 kHz)
            stt = getattr(self.checkpoint_info, "stt_config", None) or {}
            pad_left = int(stt.get("audio_silence_prefix_seconds", 0.0) * 24000)
            pad_right = int((stt.get("audio_delay_seconds", 0.0) + 1.0) * 24000)
            in_pcms = torch.nn.functional.pad(in_pcms, (pad_left, pad_right), mode="constant")
        texts: List[List[torch.Tensor]] = [[] for _ in range(B)]
        audio: List[List[torch.Tensor]] = [[] for _ in range(B)]
        done = [False] * B
        eos_id = self.text_tokenizer.eos_id() if self.text_tokenizer is not None else -1
        log("info", f"starting inference, sampling: {self.lm_gen.use_sampling}, audio temp: {self.lm_gen.temp}, "
                    f"text temp: {self.lm_gen.temp_text}")
        t0, steps = time.time(), 0
        for n, (codes, input_over) in enumerate(self._user_codes(in_pcms, lambda: all(done))):
            if n == 0:
                # the first slice of codes is stepped twice, otherwise the transformer only ever sees the initial tokens in
                # its place (run_inference.py:160-166)
                primed = self.lm_gen.step(codes)
                assert primed is None or max(lm.delays) == 0
            tokens = self.lm_gen.step(codes)
            if tokens is None:
                continue
            assert tokens.shape[1] == lm.dep_q + 1
            text = tokens[:, 0].cpu()
            if lm.dep_q == 0:                                   # ASR-style model: text only (run_inference.py:199-204)
                if self.text_tokenizer is not None and int(text[0]) not in (0, 3):
                    self.on_token(self.text_tokenizer.id_to_piece(int(text[0])).replace("\u2581", " "))
                for b in range(B):
                    texts[b].append(text[b])
                steps += 1
                continue
            pcm = self.mimi.decode(tokens[:, 1:]).cpu()
            first_live = not done[0]
            for b in range(B):
                if done[b]:
                    continue
                if int(text[b]) == eos_id:                      # run_inference.py:178-186
                    if input_over:
                        done[b] = True
                    else:
                        log("warning", "EOS sampled too early.")
                texts[b].append(text[b])
                audio[b].append(pcm[b])
            if first_live and self.text_tokenizer is not None and int(text[0]) not in (0, 3):
                self.on_token(self.text_tokenizer.id_to_piece(int(text[0])).replace("\u2581", " "))
            steps += 1
            if max_steps is not None and steps >= max_steps:
                break
        if steps:
            dt = time.time() - t0
            log("info", f"processed {steps} steps in {dt:.0f}s, {1000 * dt / steps:.2f}ms/step")
        return [(torch.cat(t, dim=0) if t else torch.zeros(0, dtype=torch.long),
                 torch.cat(a, dim=1) if a else torch.zeros(self.mimi.channels, 0)) for t, a in zip(texts, audio)]


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
