"""Import alias for the package directory `med-ts-llm_amd/` (a hyphen is not a valid Python identifier).

`import med_ts_llm_amd` executes med-ts-llm_amd/__init__.py as this module and points __path__ at that
directory, so `med_ts_llm_amd.models`, `.tasks`, `.hip`, `.utils` resolve to the files inside it.
"""
import os as _os

_pkg_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "med-ts-llm_amd")
__path__ = [_pkg_dir]
__package__ = "med_ts_llm_amd"
_init = _os.path.join(_pkg_dir, "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
del _f, _init
