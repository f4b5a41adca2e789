"""Hyper-parameters of the two models on the hot path, as plain dataclasses.

The defaults are the values the reference hard-codes for Mimi and Moshi-7B
(reference: moshi/moshi/models/loaders.py:38-88 `_seanet_kwargs`, `_quantizer_kwargs`,
`_transformer_kwargs`, `_mimi_config`; :90-119 `_lm_kwargs`).  Smaller instances of the same
architecture are used by the tests (oracle-sized).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List


@dataclass
class MimiConfig:
    sample_rate: int = 24000
    frame_rate: float = 12.5
    channels: int = 1
    dimension: int = 512
    n_filters: int = 64
    ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])  # decoder order
    kernel_size: int = 7
    last_kernel_size: int = 3
    residual_kernel_size: int = 3
    compress: int = 2
    # encoder/decoder transformer
    tr_d_model: int = 512
    tr_num_heads: int = 8
    tr_num_layers: int = 8
    tr_dim_feedforward: int = 2048
    tr_context: int = 250
    tr_max_period: float = 10000.0
    tr_layer_scale: float = 0.01
    # split residual vector quantiser
    q_dimension: int = 256
    q_bins: int = 2048
    q_n_q: int = 32
    q_n_q_semantic: int = 1

    @property
    def hop_length(self) -> int:
        h = 1
        for r in self.ratios:
            h *= r
        return h

    @property
    def frame_size(self) -> int:
        return int(self.sample_rate / self.frame_rate)

    @property
    def encoder_frame_rate(self) -> float:
        return self.sample_rate / self.hop_length

    @property
    def resample_stride(self) -> int:
        s = self.encoder_frame_rate / self.frame_rate
        assert s == int(s), "only integer resampling strides are supported"
        return int(s)

    def reference_kwargs(self) -> dict:
        """The `mimi_config` dict the reference's `loaders.get_mimi` accepts (used only by the
        golden-vector generator, which imports the reference)."""
        seanet = {
            "channels": self.channels, "dimension": self.dimension, "causal": True,
            "n_filters": self.n_filters, "n_residual_layers": 1, "activation": "ELU",
            "compress": self.compress, "dilation_base": 2, "disable_norm_outer_blocks": 0,
            "kernel_size": self.kernel_size, "residual_kernel_size": self.residual_kernel_size,
            "last_kernel_size": self.last_kernel_size, "norm": "none", "pad_mode": "constant",
            "ratios": list(self.ratios), "true_skip": True,
        }
        quantizer = {
            "dimension": self.q_dimension, "n_q": self.q_n_q, "bins": self.q_bins,
            "input_dimension": self.dimension, "output_dimension": self.dimension,
        }
        transformer = {
            "d_model": self.tr_d_model, "num_heads": self.tr_num_heads, "num_layers": self.tr_num_layers,
            "causal": True, "layer_scale": self.tr_layer_scale, "context": self.tr_context,
            "conv_layout": True, "max_period": self.tr_max_period, "gating": "none", "norm": "layer_norm",
            "positional_embedding": "rope", "dim_feedforward": self.tr_dim_feedforward,
            "input_dimension": self.dimension, "output_dimensions": [self.dimension],
        }
        return {"sample_rate": self.sample_rate, "channels": self.channels, "frame_rate": self.frame_rate,
                "seanet": seanet, "quantizer": quantizer, "transformer": transformer}


def tiny_mimi_config() -> MimiConfig:
    """A Mimi small enough for the numpy oracle and the kernel simulator: same architecture, frame = 96 samples."""
    return MimiConfig(sample_rate=1200, frame_rate=12.5, dimension=32, n_filters=4, ratios=[4, 3, 2, 2],
                      tr_d_model=32, tr_num_heads=2, tr_num_layers=2, tr_dim_feedforward=64, tr_context=6,
                      q_dimension=16, q_bins=48, q_n_q=5, q_n_q_semantic=1)


@dataclass
class LMConfig:
    dim: int = 4096
    num_heads: int = 32
    num_layers: int = 32
    hidden_scale: float = 4.125
    context: int = 3000
    max_period: float = 10000.0
    n_q: int = 16
    dep_q: int = 8
    card: int = 2048
    text_card: int = 32000
    existing_text_padding_id: int = 3
    existing_text_end_padding_id: int = 0   # lm.py:100,123: what `LMModel.end_of_text_padding_id` returns (a checkpoint's lm_kwargs may set it)
    depformer_dim: int = 1024
    depformer_dim_feedforward: int = int(4.125 * 1024)
    depformer_num_heads: int = 16
    depformer_num_layers: int = 6
    delays: List[int] = field(default_factory=lambda: [0, 0, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 1, 1, 1, 1, 1])
    extra_heads_num_heads: int = 0      # lm.py:101-102: linear heads on the transformer output (step_with_extra_heads)
    extra_heads_dim: int = 6
    kv_cache_dtype: str = "bf16"        # "fp8": e4m3 ring for the temporal transformer's keys / values (engine option)
    cross_attention: bool = False       # every temporal layer attends to the fuser's `cross` condition (transformer.py:727-731, 779-786)

    @staticmethod
    def _gating_hidden(dim: int, dim_feedforward: int) -> int:
        # reference: modules/gating.py:55-58
        if dim_feedforward == 4 * dim:
            return (21 * dim) // 8
        return (2 * dim_feedforward) // 3

    @property
    def ffn_hidden(self) -> int:
        return self._gating_hidden(self.dim, int(self.hidden_scale * self.dim))

    @property
    def depformer_ffn_hidden(self) -> int:
        return self._gating_hidden(self.depformer_dim, self.depformer_dim_feedforward)

    @property
    def num_codebooks(self) -> int:
        return self.n_q + 1

    @property
    def max_delay(self) -> int:
        return max(self.delays)

    def reference_kwargs(self) -> dict:
        """kwargs for the reference's `LMModel(...)` (golden generator only)."""
        return {
            "dim": self.dim, "text_card": self.text_card,
            "existing_text_padding_id": self.existing_text_padding_id, "n_q": self.n_q, "dep_q": self.dep_q,
            "card": self.card, "num_heads": self.num_heads, "num_layers": self.num_layers,
            "hidden_scale": self.hidden_scale, "causal": True, "layer_scale": None, "context": self.context,
            "max_period": self.max_period, "gating": "silu", "norm": "rms_norm_f32",
            "positional_embedding": "rope", "depformer_dim": self.depformer_dim,
            "depformer_dim_feedforward": self.depformer_dim_feedforward,
            "depformer_num_heads": self.depformer_num_heads, "depformer_num_layers": self.depformer_num_layers,
            "depformer_layer_scale": None, "depformer_multi_linear": True, "depformer_context": self.dep_q,
            "depformer_max_period": 10000, "depformer_gating": "silu", "depformer_pos_emb": "none",
            "depformer_weights_per_step": True, "delays": list(self.delays),
            **({"extra_heads_num_heads": self.extra_heads_num_heads, "extra_heads_dim": self.extra_heads_dim}
               if self.extra_heads_num_heads else {}),
            **({"cross_attention": True} if self.cross_attention else {}),
        }


def tiny_stt_config() -> LMConfig:
    """An ASR-style model at oracle size (the reference's kyutai/stt-* family): every audio codebook is input, no depformer
    (`dep_q = 0`, lm.py:218-221), the text stream runs `delays[0]` steps behind the audio, extra heads on the transformer
    output (lm.py:224-226) read with `step_with_extra_heads`."""
    return LMConfig(dim=128, num_heads=4, num_layers=2, hidden_scale=4.125, context=12, n_q=8, dep_q=0, card=64, text_card=96,
                    depformer_dim=64, depformer_dim_feedforward=int(4.125 * 64), depformer_num_heads=2, depformer_num_layers=2,
                    delays=[2] + [0] * 8, extra_heads_num_heads=2, extra_heads_dim=6)


def tiny_lm_config() -> LMConfig:
    """Same architecture as Moshi-7B at oracle size."""
    return LMConfig(dim=128, num_heads=4, num_layers=2, hidden_scale=4.125, context=12, n_q=16, dep_q=8, card=64,
                    text_card=96, depformer_dim=64, depformer_dim_feedforward=int(4.125 * 64),
                    depformer_num_heads=2, depformer_num_layers=2)
