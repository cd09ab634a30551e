"""CPU tests of the product's host side: the C-ABI library loads and exports what include/srk.h declares,
the ctypes structs mirror the header, the model classes keep the reference's constructor / state_dict /
init contract, the HIP path refuses to run without a GPU, and config 0 (SRCNN x2 on CPU) works."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

import sr_amd
from oracle import fill

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))
HEADER = open(os.path.join(ROOT, "include", "srk.h")).read()


def _declared_functions():
    return sorted(set(re.findall(r"^(?:int|long long|const char\*)\s+(srk_\w+)\s*\(", HEADER, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = sr_amd._lib.load()
    names = _declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/srk.h but not exported"
    assert set(names) == set(sr_amd._lib.LAUNCHERS) | set(sr_amd._lib.OTHER_SYMBOLS)
    assert lib.srk_version() >= 100
    assert [lib.srk_conv_tile(c) for c in (3, 48, 64, 102, 128, 192, 256, 768)] == [32, 64, 64, 128, 128, 64, 128, 128]


def _header_struct_fields(name):
    header = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)          # comments first: some contain braces
    body = re.search(r"typedef struct(?:\s+\w+)?\s*\{([^{}]*)\}\s*" + name + r"\s*;", header).group(1)
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        # "const void* x" / "int x_pitch, x_coff" / "float scale"
        m = re.match(r"(const\s+)?(void|float|int|double|long long|signed char|uint8_t|unsigned int|srk_patch_desc|srk_adam_slot|srk_adam_block)\s*(\*?)\s*(.*)", stmt)
        base, ptr, rest = m.group(2), m.group(3), m.group(4)
        base = {"long long": "q"}.get(base, base)
        for nm in rest.split(","):
            nm = nm.strip()
            is_ptr = bool(ptr) or nm.startswith("*")
            fields.append((nm.lstrip("* "), "p" if is_ptr else base[0]))
    return fields


@pytest.mark.parametrize("cname,cls", [("srk_pack_args", "PackArgs"), ("srk_conv_args", "ConvArgs"), ("srk_wgrad_args", "WgradArgs"),
                                        ("srk_wgrad_fin_args", "WgradFinArgs"), ("srk_unfold_args", "UnfoldArgs"),
                                        ("srk_to_nhwc_args", "ToNhwcArgs"), ("srk_to_nchw_args", "ToNchwArgs"),
                                        ("srk_ca_pool_args", "CaPoolArgs"), ("srk_ca_apply_args", "CaApplyArgs"),
                                        ("srk_ca_bwd_args", "CaBwdArgs"), ("srk_patch_desc", "PatchDesc"), ("srk_patch_args", "PatchArgs"),
                                        ("srk_sse_args", "SseArgs"), ("srk_ssim_args", "SsimArgs"), ("srk_l1_args", "L1Args"),
                                        ("srk_unfold_nhwc_args", "UnfoldNhwcArgs"), ("srk_fold_nhwc_args", "FoldNhwcArgs"),
                                        ("srk_chan_stats_args", "ChanStatsArgs"), ("srk_chan_apply_args", "ChanApplyArgs"),
                                        ("srk_conv_pair_args", "ConvPairArgs"), ("srk_adam_slot", "AdamSlot"),
                                        ("srk_adam_block", "AdamBlock"), ("srk_adam_args", "AdamArgs"), ("srk_rowsum_job", "RowsumJob"),
                                        ("srk_chan_finalize_args", "ChanFinalizeArgs"), ("srk_hrtail_args", "HrTailArgs")])
def test_ctypes_structs_mirror_header(cname, cls):
    want = _header_struct_fields(cname)
    st = getattr(sr_amd._lib, cls)
    kind = {ctypes.c_void_p: "p", ctypes.c_int: "i", ctypes.c_float: "f", ctypes.c_longlong: "q"}
    got = [(n, kind[t]) for n, t in st._fields_]
    assert got == want


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_state_dict_layout_and_init_match_reference(name):
    ent = MANIFEST[name]
    torch.manual_seed(0)
    m = getattr(sr_amd, ent["class"])(**ent["kwargs"])
    sd = m.state_dict()
    assert list(sd.keys()) == [k for k, _ in ent["state_dict"]]
    for k, shp in ent["state_dict"]:
        assert list(sd[k].shape) == shp
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(ent["trainable"])
    for k, ref in ent["init_seed0"].items():
        t = sd[k].double()
        got = [float(t.sum()), float(t.abs().sum())] + [float(v) for v in sd[k].flatten()[:3]]
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7, err_msg=k)
    if ent["class"] != "SRResNet":              # BatchNorm running statistics are the only buffers in the zoo
        assert not list(m.named_buffers())      # like the reference


def test_ctor_defaults_and_registry():
    assert sr_amd.models.__all__ == ['DDBPN', 'EDSR', 'RCAN', 'RDN', 'SRCNN', 'SRModel', 'SRResNet', 'WDSR']
    m = sr_amd.EDSR()
    assert (m._batch_size, m._channels, m._scale_factor) == (16, 3, 4)
    assert tuple(m.example_input_array.shape) == (16, 3, 32, 32)     # patch_size 128 // scale 4
    assert m.compute_dtype == torch.float32
    assert sr_amd.EDSR(precision="bf16").compute_dtype == torch.bfloat16
    assert sr_amd.EDSR(precision=16).compute_dtype == torch.float16
    with pytest.raises(ValueError):
        sr_amd.RDN(scale_factor=8)
    with pytest.raises(AttributeError):
        sr_amd.EDSR(losses="l1 + nope")
    with pytest.raises(NotImplementedError):
        sr_amd.EDSR(losses="lpips")
    with pytest.raises(ValueError):
        sr_amd.EDSR(optimizer="LION")
    with pytest.raises(ValueError):
        sr_amd.EDSR(losses="x*l1")


def test_hip_models_refuse_cpu():
    for cls, kw in (("EDSR", dict(n_feats=16, n_resblocks=1)), ("RCAN", dict(n_feats=16, n_resblocks=1, n_resgroups=1, reduction=4)),
                    ("WDSR", dict(n_feats=16, n_resblocks=1)), ("RDN", dict(rdn_config="A", G0=16))):
        m = getattr(sr_amd, cls)(**kw)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            m(torch.rand(1, 3, 8, 8))


def test_product_does_not_import_oracle():
    import sys
    pkg = os.path.join(ROOT, "sr-pytorch-lightning_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f"{f} imports the oracle"


def test_reference_state_dict_loads_strict():
    m = sr_amd.RCAN(n_feats=16, n_resblocks=2, n_resgroups=2, reduction=4)
    from oracle import init as OI
    sd, _ = OI.build_state_dict("RCAN", n_feats=16, n_resblocks=2, n_resgroups=2, reduction=4)
    m.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("name", ["srcnn_x2", "srcnn_x4"])
def test_srcnn_cpu_plumbing_matches_reference(name):
    """BASELINE config 0: SRCNN on CPU, against the reference's golden output and gradients."""
    ent = MANIFEST[name]
    g = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    m = sr_amd.SRCNN(**ent["kwargs"])
    fill.formula_fill_module(m)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = m(x)
    np.testing.assert_allclose(y.detach().numpy(), g["y"], atol=2e-6)
    (y * fill.formula_tensor(tuple(y.shape), 77, 1.0)).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], atol=2e-5)
    res = m.training_step({"lr": torch.rand(4, 3, 32, 32), "hr": torch.rand(4, 3, 32 * ent["kwargs"]["scale_factor"], 32 * ent["kwargs"]["scale_factor"])}, 0)
    assert set(res) == {"loss", "loss/l1"}
    opt = m.configure_optimizers()[0]
    assert isinstance(opt, torch.optim.Adam) and opt.defaults["lr"] == 1e-3


def test_adam_on_cpu_parameters_is_torch_adam():
    """sr_amd.optim.Adam subclasses torch.optim.Adam: CPU-resident models (SRCNN, the reference's CPU-runnable case) step
    through torch's implementation; the HIP launch is for GPU parameters only."""
    import sr_amd
    g = torch.Generator().manual_seed(0)
    ps = [torch.nn.Parameter(torch.rand(5, 3, generator=g)), torch.nn.Parameter(torch.rand(7, generator=g))]
    rs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt, ropt = sr_amd.optim.Adam(ps, lr=1e-2), torch.optim.Adam(rs, lr=1e-2)
    for _ in range(3):
        for p, r in zip(ps, rs):
            p.grad = torch.rand(p.shape, generator=g)
            r.grad = p.grad.clone()
        opt.step()
        ropt.step()
    for p, r in zip(ps, rs):
        assert torch.equal(p, r)
    with pytest.raises(NotImplementedError):
        sr_amd.optim.Adam(ps, amsgrad=True)
