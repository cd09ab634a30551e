"""Importable alias of the `sr-pytorch-lightning_amd/` package (a hyphen cannot be imported with `import`)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("sr-pytorch-lightning_amd")
# make `import sr_amd.models` etc. resolve to the SAME module objects
for _name, _mod in list(sys.modules.items()):
    if _name.startswith("sr-pytorch-lightning_amd."):
        sys.modules["sr_amd." + _name.split(".", 1)[1]] = _mod
sys.modules[__name__] = _pkg
