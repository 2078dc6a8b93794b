"""Model hyper-parameters of the denoiser hot path.

Flag names follow the reference's argparse (`mdgen/parsing.py:79-120`) so that a Lightning
checkpoint's ``hyper_parameters['args']`` Namespace maps 1:1 (see `ModelConfig.from_args`).
"""
from __future__ import annotations

from dataclasses import dataclass, asdict


@dataclass
class ModelConfig:
    embed_dim: int = 384
    num_layers: int = 5
    mha_heads: int = 16
    ipa_heads: int = 4
    ipa_head_dim: int = 32
    ipa_qk: int = 8
    ipa_v: int = 8
    time_multiplier: float = 100.0
    crop: int = 4
    num_frames: int = 1000
    abs_pos_emb: bool = True
    sim_condition: bool = True
    tps_condition: bool = False
    prepend_ipa: bool = True
    no_rope: bool = False

    @property
    def latent_dim(self) -> int:
        # wrapper.py:196 -- 21 for forward-sim, 28 with two-sided (TPS) conditioning
        return 28 if self.tps_condition else 21

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.mha_heads

    def to_dict(self):
        return asdict(self)

    @staticmethod
    def from_args(args) -> "ModelConfig":
        """Build from the reference's argparse Namespace (ckpt['hyper_parameters']['args'])."""
        kw = {}
        for f in ModelConfig.__dataclass_fields__:
            if hasattr(args, f):
                kw[f] = getattr(args, f)
        return ModelConfig(**kw)

    @staticmethod
    def forward_sim(num_frames=1000, crop=4) -> "ModelConfig":
        """README.md:48 tetrapeptide forward-simulation model."""
        return ModelConfig(crop=crop, num_frames=num_frames, abs_pos_emb=True, sim_condition=True)

    @staticmethod
    def atlas(num_frames=250, crop=256) -> "ModelConfig":
        """README.md:60 ATLAS model (no abs_pos_emb)."""
        return ModelConfig(crop=crop, num_frames=num_frames, abs_pos_emb=False, sim_condition=True)

    @staticmethod
    def tps(num_frames=100, crop=4) -> "ModelConfig":
        return ModelConfig(crop=crop, num_frames=num_frames, abs_pos_emb=True, sim_condition=False,
                           tps_condition=True)
