"""cmvae_amd — MI355X-native hot path of CMVAE (episodic train step + few-shot eval).

Importing this package loads libcmvae_hip.so; it fails loudly if the library is missing.
"""
from . import _lib, ops  # noqa: F401  (load the C-ABI first: no fallback)
from .model import OmniglotGMVAE, FeatureGMVAE  # noqa: F401
from . import registry  # noqa: F401  (torch.ops.cmvae_amd.* for the remaining leaf entry points)
