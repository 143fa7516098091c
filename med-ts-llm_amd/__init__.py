"""med-ts-llm_amd — MI355X-native MedTsLLM forward/backward hot path behind the reference's plugin surface.

Sub-packages
  csrc/    hand-written HIP kernels for gfx950 + the C-ABI (include/medtsllm_hip.h)
  hip/     ctypes binding of libmedtsllm_hip.so and the torch.autograd.Function wrappers
  models/  `model_lookup` registry + the drop-in `MedTsLLM` module (reference: models/__init__.py, models/medtsllm.py)
  tasks/   BaseTask-compatible trainer whose loop body equals the reference's (tasks/forecasting.py:19-30) + DP
  utils.py `dict_to_object` config object (reference: utils.py:19-39)
"""
__version__ = "0.1.0"
