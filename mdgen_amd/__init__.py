"""mdgen_amd -- MI355X-native MDGen denoising sampler (hot path only)."""
from .config import ModelConfig  # noqa: F401

__version__ = "0.1.0"
