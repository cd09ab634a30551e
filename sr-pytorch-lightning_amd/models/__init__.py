"""Model registry with the reference's names (models/__init__.py:1-21).

EDSR, RCAN, RDN, WDSR, SRResNet and DDBPN run on the HIP path; SRCNN is the CPU plumbing case.  SRGAN (adversarial
training, broken in the reference itself: SURVEY.md section 8(c)) is outside this build's scope and is not exported.
"""
from .ddbpn import DDBPN
from .edsr import EDSR
from .rcan import RCAN
from .rdn import RDN
from .srcnn import SRCNN
from .srmodel import SRModel
from .srresnet import SRResNet
from .wdsr import WDSR

__all__ = ['DDBPN', 'EDSR', 'RCAN', 'RDN', 'SRCNN', 'SRModel', 'SRResNet', 'WDSR']
