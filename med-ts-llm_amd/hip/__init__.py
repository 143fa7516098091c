from ._native import lib, available, MtlError  # noqa: F401
