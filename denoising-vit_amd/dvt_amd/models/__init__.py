"""`dvt.models` surface of the reference (dvt/models/__init__.py): same names."""
from .neural_feature_field import HashGridEncoding, HipLinear, NeuralFeatureField  # noqa: F401
from .offline_denoiser import SingleImageDenoiser  # noqa: F401
from .vit_wrapper import MODEL_LIST, PretrainedViTWrapper  # noqa: F401
from .online_denoiser import Denoiser  # noqa: F401
