"""Graph-replayed training steps follow the oracle's trajectory."""


import numpy as np


import pytest


import torch


from oracle import functional as OF, train as OT


pytestmark = pytest.mark.gpu


PREC = {torch.float16: 16, torch.bfloat16: "bf16"}


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


RCAN_KW = dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)


@pytest.mark.parametrize("cls,kw", [("RCAN", dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)),
                                    ("EDSR", dict(n_feats=64, n_resblocks=4, res_scale=0.1, scale_factor=2)),
                                    # weight-normed convs: the effective weights are non-leaf tensors (ops.WeightNormGroup); a
                                    # reference to one that outlives its eager step used to crash the capture of a later step
                                    ("WDSR", dict(type="B", n_feats=64, n_resblocks=3, scale_factor=2)),
                                    ("WDSR", dict(type="A", n_feats=32, n_resblocks=2, scale_factor=2))])
def test_graph_replayed_steps_follow_the_oracle_trajectory(A, cls, kw):
    """Trainer.fit (3 eager steps, then hipGraph replays of the pair-kernel step) against the ORACLE's Adam trajectory on the
    same batches (srmodel.py:145-171): the loss of every step, computed from weights that all earlier steps produced."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = getattr(A, cls)(precision="bf16", **kw)
    om = OT.OracleModel(cls, **kw)
    om.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    steps = 8
    data = [T.synthetic_batch(16, 3, 48, 2, 500 + i, "cpu") for i in range(steps)]
    # learnable targets (HR = bilinear upsampling of LR) so that the loss moves by far more than the comparison tolerance
    for b in data:
        b["hr"] = torch.nn.functional.interpolate(b["lr"], scale_factor=2, mode="bilinear", align_corners=False)
    opt = om.configure_optimizers()[0]
    ref = []
    for b in data:
        opt.zero_grad()
        loss = om.training_step(b)["loss"]
        loss.backward()
        opt.step()
        ref.append(float(loss))
    tr = T.Trainer(device="cuda", use_graph=True)
    tr.fit(m, iter(data))
    torch.cuda.synchronize()
    assert tr.graphed is not None and tr.graphed.graphs is not None and not tr.graphed.failed
    got = tr.losses
    assert len(got) == steps
    assert ref[0] - ref[-1] > 0.05 * ref[0], f"the oracle's loss should fall on learnable data: {ref}"
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-3)
