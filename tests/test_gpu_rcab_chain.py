"""RCAN's 64-feature RCAB chain (two convs per launch with the channel attention inside, models/rcan.py:10-74) against the oracle."""


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


@pytest.mark.parametrize("dt,min_psnr,min_cos", [(torch.bfloat16, 50.0, 0.99), (torch.float16, 62.0, 0.999)])
def test_rcan_64_feature_chain_16bit_vs_oracle(A, dt, min_psnr, min_cos):
    """BASELINE config 3's own shape per launch: RCAN with 64 features at 16 x 3 x 48 x 48 (one 14x14 tile per CU, the pair
    kernel with pooling, `ca_mode 2` forward and `ca_mode 1` backward through `ops.rcab_chain`), forward PSNR and the cosine
    of every sizeable parameter gradient against the fp32 oracle (rcan.py:33-74), plus the channel-attention parameters'
    gradients (conv_du: 64 -> 4 -> 64), which only the fused backward produces."""
    from sr_amd import ops
    torch.manual_seed(0)
    m = A.RCAN(precision=PREC[dt], **RCAN_KW)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    for k in trainable:
        sd[k].requires_grad_(True)
    m = m.cuda()
    gen = torch.Generator().manual_seed(777)
    x = torch.rand(16, 3, 48, 48, generator=gen)
    hr = torch.rand(16, 3, 96, 96, generator=gen)
    y_ref = OF.forward("RCAN", sd, x, **RCAN_KW)
    torch.nn.functional.l1_loss(y_ref, hr).backward()
    before = list(ops.PAIR_LAUNCHES)
    y = m(x.cuda())
    loss = torch.nn.functional.l1_loss(y, hr.cuda())
    (loss * 1024.0).backward()                      # loss scaling keeps fp16 gradients out of the subnormals
    torch.cuda.synchronize()
    took = [a - b for a, b in zip(ops.PAIR_LAUNCHES, before)]
    # 2 groups x 3 RCABs: forward = per group 1 plain + 2 with the previous block's CALayer forward (ca_mode 2);
    # backward = 6 launches with the CALayer backward on the way in (ca_mode 1)
    assert took[2] == 4 and took[1] == 6 and took[0] >= 2, f"pair launches by ca_mode: {took}"
    mse = float(((y.detach().cpu().double() - y_ref.detach().double()) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    assert psnr > min_psnr, f"RCAN 64 {dt}: PSNR(build, oracle) = {psnr:.1f} dB"
    assert abs(float(loss) - float(torch.nn.functional.l1_loss(y_ref, hr))) < 2e-3
    params = dict(m.named_parameters())
    worst = 1.0
    for k in sorted(trainable):
        ref = sd[k].grad.double().flatten()
        if ref.numel() < 256:
            continue
        got = params[k].grad.cpu().double().flatten() / 1024.0
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > min_cos, f"RCAN 64 {dt} grad {k}: cosine {cos:.5f}"
        ratio = float(got.norm() / (ref.norm() + 1e-30))
        assert 0.9 < ratio < 1.1, f"RCAN 64 {dt} grad {k}: norm ratio {ratio:.3f}"
    ca_keys = [k for k in trainable if "conv_du" in k and k.endswith("weight")]
    assert len(ca_keys) == 12
    print(f"RCAN 64 {dt}: PSNR {psnr:.1f} dB, worst gradient cosine {worst:.5f}, pair launches {took}")
