"""Import alias: the package sources live in ``dynamic-asr-eval_amd/`` (a directory name Python cannot import
directly because of the hyphens).  ``import dynamic_asr_eval_amd`` resolves every submodule from there."""
import os as _os

_src = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "dynamic-asr-eval_amd")
__path__.insert(0, _src)
with open(_os.path.join(_src, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_src, "__init__.py"), "exec"))
del _os, _f
