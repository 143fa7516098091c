"""Config object + small helpers with the reference's semantics (reference: utils.py:7-39,86-95)."""
import random
from copy import deepcopy
from datetime import datetime

import torch


def get_run_id(debug=False):
    run_id = datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    return ("DEBUG-" + run_id) if debug else run_id


def set_seed(seed):
    random.seed(seed)
    torch.manual_seed(seed)


class dict_to_object(object):
    """Nested dict -> attribute access, `.get(k, default)`, `k in cfg`, `cfg[k]`, `.copy()`, `.to_dict()`.

    Copied from R:utils.py:19-39 (14 lines): it IS the boundary type — the reference's trainer, model constructor and loggers
    duck-type exactly these seven members on the config they are handed, so a drop-in has to be the same class."""

    def __init__(self, d):
        self.__dict__ = {k: dict_to_object(v) if isinstance(v, dict) else v for k, v in d.items()}

    def to_dict(self):
        return {k: v.to_dict() if isinstance(v, dict_to_object) else v for k, v in self.__dict__.items()}

    def get(self, key, default=None):
        return self.__dict__.get(key, default)

    def __getitem__(self, key):
        return self.__dict__[key]

    def __contains__(self, key):
        return key in self.__dict__

    def __repr__(self):
        return str(self.__dict__)

    def copy(self):
        return deepcopy(self)


def get_dtype(dtype_name):
    if dtype_name in ("bfloat16", "bf16"):
        return torch.bfloat16
    if dtype_name in ("float16", "half", "fp16", "16", 16):
        return torch.float16
    if dtype_name in ("float32", "float", "fp32", "32", 32, "mixed"):
        return torch.float32
    raise ValueError(f"Invalid dtype selection: {dtype_name}")
