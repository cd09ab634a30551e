"""Pins the CPU oracle against every golden vector produced from the reference.

Fixtures: tests/golden/*.npz + manifest.json (made by tests/golden/generate_golden.py
from /root/reference).  Tolerances: the oracle and the reference run the same torch
CPU kernels in the same order for the forward pass, so outputs must agree to
float32 round-off (rtol 1e-5 of the output range); gradients likewise.
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

from oracle import fill, functional as OF, init as OI, train as OT

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


def _close(a, b, tol=2e-5):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(1e-6, float(np.abs(b).max()))
    err = float(np.abs(a - b).max()) / scale
    assert err < tol, f"rel-to-range err {err:.3e} >= {tol}"


def _filled_sd(cls, kw):
    sd, trainable = OI.build_state_dict(cls, **kw)
    fill.formula_fill_state_dict(sd, trainable)
    for k in trainable:
        sd[k].requires_grad_(True)
    return sd, trainable


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_manifest_keys_shapes_and_init(name):
    """state_dict key order/shapes and seed-0 torch-default init match the reference."""
    ent = MANIFEST[name]
    torch.manual_seed(0)
    sd, trainable = OI.build_state_dict(ent["class"], **ent["kwargs"])
    assert list(sd.keys()) == [k for k, _ in ent["state_dict"]]
    for k, shp in ent["state_dict"]:
        assert list(sd[k].shape) == shp, k
    assert sorted(trainable) == sorted(ent["trainable"])
    assert sum(sd[k].numel() for k in trainable) == ent["n_params_trainable"]
    for k, ref in ent["init_seed0"].items():
        t = sd[k].double()
        got = [float(t.sum()), float(t.abs().sum())] + [float(v) for v in sd[k].flatten()[:3]]
        np.testing.assert_allclose(got, ref, rtol=1e-6, atol=1e-7, err_msg=k)


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_model_forward_backward(name):
    ent = MANIFEST[name]
    g = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    sd, trainable = _filled_sd(ent["class"], ent["kwargs"])
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    # the fixture input is the formula input: check the restated generator too
    _close(fill.formula_input(tuple(g["x"].shape)).numpy(), g["x"], 1e-7)
    y = OF.forward(ent["class"], sd, x, **ent["kwargs"])
    _close(y.detach().numpy(), g["y"])
    t = fill.formula_tensor(tuple(y.shape), 77, 1.0)
    (y * t).sum().backward()
    _close(x.grad.numpy(), g["dx"], 1e-4 if ent["n_params_trainable"] < 1_000_000 else 5e-3)
    names = [str(n) for n in g["grad_names"]]
    sums = g["grad_sums"]
    # full-depth nets amplify float32 round-off through 30-400 ReLU layers (the same torch ops in a
    # different association: e.g. WDSR's w = v*(g/|v|) here vs torch._weight_norm there)
    rt = 2e-4 if ent["n_params_trainable"] < 1_000_000 else 1e-2
    for n, s in zip(names, sums):
        gr = sd[n].grad.double().flatten()
        got = np.array([float(gr.sum()), float(gr.abs().sum()), float((gr * gr).sum())])
        # abs-sum and square-sum are well conditioned; the plain sum can cancel
        np.testing.assert_allclose(got[1:], s[1:], rtol=rt, err_msg=n)
        assert abs(got[0] - s[0]) <= rt * max(s[1], 1e-12), n
        if "g:" + n in g:
            _close(sd[n].grad.numpy(), g["g:" + n], 1e-4)
    if "y_eval" in g:          # BatchNorm models: eval-mode output after the ONE training-mode forward above, and the buffers
        with torch.no_grad():
            ye = OF.forward(ent["class"], sd, torch.from_numpy(g["x"]), training=False, **ent["kwargs"])
        _close(ye.numpy(), g["y_eval"])
        for n, s3 in zip([str(v) for v in g["buffer_names"]], g["buffer_sums"]):
            b = sd[n].double().flatten()
            np.testing.assert_allclose([float(b.sum()), float(b.abs().sum()), float((b * b).sum())], s3, rtol=1e-5, atol=1e-9, err_msg=n)


BLOCKS = {
    "block_resblock64_rs0.1": lambda sd, x: OF.res_block(sd, "B", x, 0.1),
    "block_resblock256": lambda sd, x: OF.res_block(sd, "B", x, 1.0),
    "block_upscale4_64": lambda sd, x: OF.upscale_block(sd, "B", x, 4, 64),
    "block_upscale3_64": lambda sd, x: OF.upscale_block(sd, "B", x, 3, 64),
    "block_upscale2_64": lambda sd, x: OF.upscale_block(sd, "B", x, 2, 64),
    "block_calayer64_r16": lambda sd, x: OF.ca_layer(sd, "B", x),
    "block_rcab64": lambda sd, x: OF.rcab(sd, "B", x),
    "block_resgroup64_b2": lambda sd, x: OF.residual_group(sd, "B", x, 2),
    "block_rdb_64_64_8": lambda sd, x: OF.rdb(sd, "B", x, 8),
    "block_rdb_16_8_3": lambda sd, x: OF.rdb(sd, "B", x, 3),
    "block_wdsr_a128": lambda sd, x: OF.wdsr_block(sd, "B", x, "A", 1),
    "block_wdsr_b128": lambda sd, x: OF.wdsr_block(sd, "B", x, "B", 1),
}


def _block_sd(name):
    """Rebuild the block's parameter names/shapes as the reference module lays them out."""
    from collections import OrderedDict
    sd = OrderedDict()

    def conv(p, ci, co, k, wn=False):
        if wn:
            sd[p + ".bias"] = torch.zeros(co)
            sd[p + ".weight_g"] = torch.zeros(co, 1, 1, 1)
            sd[p + ".weight_v"] = torch.zeros(co, ci, k, k)
        else:
            sd[p + ".weight"] = torch.zeros(co, ci, k, k)
            sd[p + ".bias"] = torch.zeros(co)
    if name.startswith("block_resblock"):
        f = 64 if "64" in name else 256
        conv("B.body.0", f, f, 3); conv("B.body.2", f, f, 3)
    elif name.startswith("block_upscale"):
        s = int(name.split("upscale")[1][0]); r = 2 if s % 2 == 0 else 3
        for i in range({2: 1, 3: 1, 4: 2}[s]):
            conv(f"B.{2*i}", 64, 64 * r * r, 3)
    elif name == "block_calayer64_r16":
        conv("B.conv_du.0", 64, 4, 1); conv("B.conv_du.2", 4, 64, 1)
    elif name == "block_rcab64":
        conv("B.body.0", 64, 64, 3); conv("B.body.2", 64, 64, 3)
        conv("B.body.3.conv_du.0", 64, 4, 1); conv("B.body.3.conv_du.2", 4, 64, 1)
    elif name == "block_resgroup64_b2":
        for b in range(2):
            conv(f"B.body.{b}.body.0", 64, 64, 3); conv(f"B.body.{b}.body.2", 64, 64, 3)
            conv(f"B.body.{b}.body.3.conv_du.0", 64, 4, 1); conv(f"B.body.{b}.body.3.conv_du.2", 4, 64, 1)
        conv("B.body.2", 64, 64, 3)
    elif name.startswith("block_rdb"):
        g0, g, c = (64, 64, 8) if "64_64_8" in name else (16, 8, 3)
        for i in range(c):
            conv(f"B.convs.{i}.conv.0", g0 + i * g, g, 3)
        conv("B.LFF", g0 + c * g, g0, 1)
    elif name == "block_wdsr_a128":
        conv("B.body.0", 128, 512, 3, True); conv("B.body.2", 512, 128, 3, True)
    elif name == "block_wdsr_b128":
        conv("B.body.0", 128, 768, 1, True); conv("B.body.2", 768, 102, 1, True); conv("B.body.3", 102, 128, 3, True)
    return sd


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_forward_backward(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = _block_sd(name)
    # named_parameters() of the bare block has no "B." prefix but the same order
    fill.formula_fill_state_dict(sd, set(sd))
    for v in sd.values():
        v.requires_grad_(True)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = BLOCKS[name](sd, x)
    _close(y.detach().numpy(), g["y"])
    (y * fill.formula_tensor(tuple(y.shape), 77, 1.0)).sum().backward()
    _close(x.grad.numpy(), g["dx"], 1e-4)
    for n, s in zip([str(v) for v in g["grad_names"]], g["grad_sums"]):
        gr = sd["B." + n].grad.double().flatten()
        np.testing.assert_allclose([float(gr.abs().sum()), float((gr * gr).sum())], s[1:], rtol=2e-4, err_msg=n)


def test_meanshift():
    g = np.load(os.path.join(GOLDEN, "block_meanshift.npz"))
    sd, _ = OI.build_state_dict("EDSR", n_feats=16, n_resblocks=1)
    for k in ("weight", "bias"):
        np.testing.assert_array_equal(sd[f"sub_mean.{k}"].numpy(), g[f"sub_{k}"])
        np.testing.assert_array_equal(sd[f"add_mean.{k}"].numpy(), g[f"add_{k}"])
    x = torch.from_numpy(g["x"])
    np.testing.assert_allclose(OF.mean_shift(sd, "sub_mean", x).numpy(), g["y_sub"], atol=1e-7)
    np.testing.assert_allclose(OF.mean_shift(sd, "add_mean", x).numpy(), g["y_add"], atol=1e-7)


@pytest.mark.parametrize("tag,losses,opt", [("l1_adam", "l1", "ADAM"), ("l2_sgd", "0.5*l2 + 0.5*l1", "SGD")])
def test_training_trajectory(tag, losses, opt):
    """3 optimizer steps through the reference's training_step/configure_optimizers."""
    g = np.load(os.path.join(GOLDEN, f"traj_edsr_f16_b2_x4_{tag}.npz"))
    m = OT.OracleModel("EDSR", losses=losses, optimizer=opt, optimizer_params=["lr=1e-2"],
                       n_feats=16, n_resblocks=2, res_scale=0.1, scale_factor=4)
    fill.formula_fill_state_dict(m.sd, m.trainable)
    optim = m.configure_optimizers()[0]
    ref_defaults = json.loads(str(g["opt_defaults"]))
    assert optim.defaults["lr"] == ref_defaults["lr"] == 1e-3      # user lr dropped, as in the reference
    for step in range(3):
        batch = {"lr": fill.formula_input((2, 3, 8, 8), k=2000 + step),
                 "hr": fill.formula_input((2, 3, 32, 32), k=3000 + step), "path": ["a", "b"]}
        optim.zero_grad()
        res = m.training_step(batch, step)
        assert sorted(res.keys()) == [str(k) for k in g[f"keys{step}"]]
        got = [float(res["loss"])] + [float(v) for k, v in sorted(res.items()) if k != "loss"]
        np.testing.assert_allclose(got, g["losses"][step], rtol=1e-5)
        res["loss"].backward()
        optim.step()
    for k in m.sd:
        _close(m.sd[k].detach().numpy(), g["w:" + k], 1e-5)


def test_loss_parser_errors():
    with pytest.raises(AttributeError):
        OT.parse_losses("l1 + nope")
    with pytest.raises(ValueError):
        OT.parse_losses("x*l1")
    assert OT.parse_losses("0.5 * L1 + mse") == [("l1", 0.5), ("mse", 1.0)]
    with pytest.raises(ValueError):
        OT.make_optimizer([], "LION")
