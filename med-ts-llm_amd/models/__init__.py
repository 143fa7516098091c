"""Model registry with the reference's surface (R:models/__init__.py:10-18): `model_lookup[config.model](config, dataset)`.

Only the MedTsLLM hot path is built here; the reference's baseline models are out of scope.
"""
from .medtsllm import MedTsLLM

model_lookup = {
    "timellm": MedTsLLM,
    "medtsllm": MedTsLLM,
}
