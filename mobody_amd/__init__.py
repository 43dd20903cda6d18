"""Importable alias of the product package.

The product lives in `mobody-model-based-off-dynamics-offline-reinforcement-learning_amd/`
(the directory name the project mandates, which is not a valid Python identifier); this
stub makes it importable as `mobody_amd` by pointing the package search path there.
"""
import os as _os

_PKG_DIR = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                         "mobody-model-based-off-dynamics-offline-reinforcement-learning_amd")
__path__ = [_PKG_DIR]
with open(_os.path.join(_PKG_DIR, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_PKG_DIR, "__init__.py"), "exec"))
