"""sr-pytorch-lightning_amd: the MI355X-native convolutional hot path of george-gca/sr-pytorch-lightning.

The directory name is not a valid Python identifier; import it as `sr_amd` (the alias module at
the repository root) or with `importlib.import_module("sr-pytorch-lightning_amd")`.
"""
from . import _lib, ops, optim, models, data  # noqa: F401
from .models import DDBPN, EDSR, RCAN, RDN, SRCNN, SRModel, SRResNet, WDSR  # noqa: F401

__all__ = ["_lib", "ops", "optim", "models", "data", "DDBPN", "EDSR", "RCAN", "RDN", "SRCNN", "SRModel", "SRResNet", "WDSR"]
