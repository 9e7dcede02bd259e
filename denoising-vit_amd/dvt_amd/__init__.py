"""dvt_amd -- MI355X-native DVT stage-1 denoising (hash-grid field + fused fit + ViT
extractor) behind the reference's `dvt.models` API.  See DESIGN.md."""
from . import _lib  # noqa: F401
from . import vit  # noqa: F401  (registers the ViT entry points before the first library load)
from . import s2  # noqa: F401  (stage-2 entry points)
from .fit import FitEngine, FitSettings  # noqa: F401
from . import models  # noqa: F401

__all__ = ["FitEngine", "FitSettings", "models", "vit"]
