"""Every epilogue variant of the weight-stationary 3x3 kernel (conv_ws_kernel), 16-bit types, through srk_conv2d:
plain / ReLU (quad-transposed stores), residual OR mask (the prefetch variant, also with Cin = 16), residual AND mask
(plain variant with in-phase loads), ragged tiles and widths that are not a multiple of the 4-pixel quad.
Reference: float64 conv of the SAME 16-bit-rounded operands, epilogue in the order of include/srk.h
(bias; relu; *scale; +res; mask; store), rounded once to the storage type."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DT = [torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def A():
    import sr_amd
    return sr_amd


@pytest.mark.parametrize("dt", DT, ids=["bf16", "f16"])
@pytest.mark.parametrize("cin", [64, 16])
@pytest.mark.parametrize("n,h,w", [(2, 48, 48), (1, 17, 50), (3, 5, 3)])
@pytest.mark.parametrize("variant", ["plain", "relu", "res", "res_scale", "mask0", "mask16", "res_mask"])
def test_ws_epilogue_variants(A, dt, cin, n, h, w, variant):
    g = torch.Generator().manual_seed(hash((cin, n, h, w, variant)) & 0xffff)
    cout = 64
    x = ((torch.rand(n, h, w, cin, generator=g) - 0.5) * 2).to(dt).cuda()
    wt = ((torch.rand(cout, cin, 3, 3, generator=g) - 0.5) * (2.0 / (cin * 9) ** 0.5)).cuda()
    b = ((torch.rand(cout, generator=g) - 0.5) * 0.2).cuda()
    res = ((torch.rand(n, h, w, cout, generator=g) - 0.5) * 2).to(dt).cuda() if "res" in variant else None
    mask = torch.relu((torch.rand(n, h, w, cout, generator=g) - 0.5)).to(dt).cuda() if "mask" in variant else None
    mask_from = 16 if variant == "mask16" else 0
    relu = variant == "relu"
    scale = 0.1 if variant == "res_scale" else 1.0
    pk = A.ops.pack_conv(torch.nn.Parameter(wt), torch.nn.Parameter(b), dt)
    out = torch.full((n, h, w, cout), 7.0, dtype=dt, device="cuda")
    A.ops.conv_raw(x, pk, N=n, H=h, W=w, Cin=cin, Cout=cout, out=out, relu=relu, scale=scale, res=res, mask=mask,
                   mask_from=mask_from)
    torch.cuda.synchronize()

    wq = wt.to(dt).double().cpu()                       # the packed weights are rounded to the compute type
    ref = F.conv2d(x.double().cpu().permute(0, 3, 1, 2), wq, b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    if relu:
        ref = ref.clamp_min(0)
    ref = ref * scale
    if res is not None:
        ref = ref + res.double().cpu()
    if mask is not None:
        keep = mask.double().cpu() > 0
        keep[..., :mask_from] = True
        ref = torch.where(keep, ref, torch.zeros_like(ref))
        assert bool((out.cpu()[~keep] == 0).all()), "masked elements are exactly zero"
    got = out.double().cpu()
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11          # half an ulp of the storage type, relative
    err = (got - ref).abs()
    tol = eps * ref.abs() * 1.5 + 4e-3 * eps * 256 / 256 + 1e-3 * float(ref.abs().max()) * (1 if dt == torch.bfloat16 else 0.1)
    assert bool((err <= tol).all()), f"max err {float(err.max()):.3e} (max |ref| {float(ref.abs().max()):.3f})"


@pytest.mark.parametrize("dt", DT, ids=["bf16", "f16"])
@pytest.mark.parametrize("cin,cout", [(128, 768), (768, 112), (32, 64), (576, 64), (64, 64), (16, 16)])
@pytest.mark.parametrize("n,h,w", [(2, 13, 7), (1, 48, 48), (1, 1, 1)])
def test_wgrad_1x1_slab_kernel(A, dt, cin, cout, n, h, w):
    """dW / db of a 1x1 conv through srk_conv2d_wgrad (the slab-mode GEMM kernel, both tile shapes, ragged 64-pixel
    K tiles, channel counts that do not fill the 64-channel blocks) against float64 on the same rounded operands."""
    g = torch.Generator().manual_seed(cin * 7 + cout + h)
    x = ((torch.rand(n, h, w, cin, generator=g) - 0.5) * 2).to(dt).cuda()
    dy = ((torch.rand(n, h, w, cout, generator=g) - 0.5) * 2).to(dt).cuda()
    dw, db = A.ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=cin, Cout=cout, k=1, w_shape=(cout, cin, 1, 1))
    torch.cuda.synchronize()
    xr, dr = x.double().cpu().reshape(-1, cin), dy.double().cpu().reshape(-1, cout)
    ref_w = (dr.t() @ xr).reshape(cout, cin, 1, 1)
    ref_b = dr.sum(0)
    tol = 2e-5 * max(1.0, float(ref_w.abs().max()))          # fp32 accumulation of exact 16-bit products
    assert float((dw.double().cpu() - ref_w).abs().max()) <= tol
    assert float((db.double().cpu() - ref_b).abs().max()) <= 2e-5 * max(1.0, float(ref_b.abs().max()))
