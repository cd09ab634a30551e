"""Model registry with the reference's names (models/__init__.py:1-21).

EDSR, RCAN, RDN, WDSR run on the HIP hot path; SRCNN is the CPU plumbing case.  DDBPN, SRGAN and
SRResNet are outside this build's scope and are not exported.
"""
from .edsr import EDSR
from .rcan import RCAN
from .rdn import RDN
from .srcnn import SRCNN
from .srmodel import SRModel
from .wdsr import WDSR

__all__ = ['EDSR', 'RCAN', 'RDN', 'SRCNN', 'SRModel', 'WDSR']
