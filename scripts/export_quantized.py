"""Write a quantised Moshi checkpoint (the reference's scripts/export_quantized.py without the Hugging Face hub):

    python scripts/export_quantized.py model.safetensors model.q8.safetensors [--format int8|fp8] [--config config.json]

int8 = the reference's `quantize=True` storage (`weight` int8 + `weight_scb`, utils/quantize.py); fp8 = e4m3fn `weight` +
`weight_scale` for the fp8 MFMA path.  `--config`: the model's config.json when it is not Moshi-7B (needed to split the fused
per-step attention projections of released checkpoints).
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--format", choices=["int8", "fp8"], default="int8")
    ap.add_argument("--config", default=None)
    args = ap.parse_args()
    from moshi_amd import loaders
    lm_kwargs = None
    if args.config:
        lm_kwargs = json.loads(Path(args.config).read_text())
        for k in ("moshi_name", "mimi_name", "tokenizer_name", "model_type", "lm_gen_config", "mimi_config", "tts_config",
                  "stt_config", "model_id", "lora_name"):
            lm_kwargs.pop(k, None)
    info = loaders.export_quantized(args.src, args.dst, args.format, lm_kwargs)
    print(f"{args.dst}: {info['tensors']} tensors, {info['quantized']} quantised, {info['bytes'] / 1e9:.2f} GB")


if __name__ == "__main__":
    main()
