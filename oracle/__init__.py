"""CPU oracle for the SR convolutional hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch-CPU / numpy restatement of the reference's
forward/backward arithmetic for `models/{common,edsr,rcan,rdn,wdsr,srcnn}.py`
and the `SRModel.training_step` caller (george-gca/sr-pytorch-lightning).
Every function cites the reference file:line it follows.

Who may import it: `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` -- as the checker / the timed CPU baseline, never as the
thing shipped.  Nothing under `sr-pytorch-lightning_amd/` imports it, and the
product raises when its HIP library is missing instead of falling back here.

Pinning: `tests/golden/*.npz` were produced by running the REFERENCE's own
modules in the build container (`tests/golden/generate_golden.py`);
`tests/test_oracle_golden.py` checks this package against every one of them.
PSNR/SSIM (`oracle.metrics`) restate piq's published defaults but piq is not
installable here: that part is "parity unpinned" (see DESIGN.md).
"""
from . import fill, functional, init, metrics, conv_np, train, data  # noqa: F401
