#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); nothing here is
imported by the tests, the bench or the product.  The fixtures it writes are
data only: inputs, outputs, gradients, checksums and key/shape manifests.

How the reference is imported: `models/srmodel.py`
imports kornia / piq / lightning / torchvision / torch_optimizer /
robust_loss_pytorch and the reference's own `losses` package at module scope
and none of them is installed here.  A `sys.meta_path` finder supplies empty
stand-in modules for exactly those roots, with
`lightning.pytorch.LightningModule := torch.nn.Module + no-op bookkeeping`.
Every FLOP in `forward()` / `training_step()` (l1 loss) / Adam is the
reference's own module graph running on real torch CPU ops; the stand-ins are
only touched by base-class bookkeeping and by loss/metric *tables*.

Weights are "formula filled" (see `formula_fill`) so that a fixture never has
to store a state_dict: both sides rebuild identical weights from the flat
index.  Usage:  python tests/golden/generate_golden.py
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("SR_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
_STUB_ROOTS = ("kornia", "piq", "lightning", "torchvision", "torch_optimizer",
               "robust_loss_pytorch", "losses")


class _Anything:
    """Callable / attribute sink used for every name pulled from a stand-in."""
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _LightningModule(nn.Module):
    """nn.Module + the bookkeeping calls SRModel.__init__ makes (no arithmetic)."""
    def save_hyperparameters(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device("cpu")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        if name == "LightningModule":
            return _LightningModule
        return _Anything


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import models  # noqa: the reference's package
    return models


# --------------------------------------------------------------------------
# deterministic, RNG-free tensors (restated identically in oracle/fill.py)
# --------------------------------------------------------------------------
def formula_tensor(shape, k, amp):
    n = int(np.prod(shape)) if len(shape) else 1
    v = amp * np.sin(0.37 * np.arange(n, dtype=np.float64) + 0.61 * k)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def formula_fill(module):
    """Overwrite every trainable parameter, in state_dict order."""
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            if not p.requires_grad:
                continue  # frozen MeanShift keeps its constructor values
            if name.endswith("weight_g"):
                p.copy_(1.0 + 0.25 * formula_tensor(p.shape, k, 1.0))
            elif p.dim() > 1:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(formula_tensor(p.shape, k, 1.0 / np.sqrt(fan_in)))
            else:
                p.copy_(formula_tensor(p.shape, k, 0.1))


def formula_input(shape, k=1000):
    # in [0,1] like a normalised image
    return 0.5 + 0.5 * formula_tensor(shape, k, 1.0)


def grad_summary(t):
    t = t.detach().double().flatten()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


def run_case(model, x, with_grads=True, full_grads=False, with_eval=False):
    """y = model(x); L = sum(y * t) with a formula target t -> grads.  with_eval (models with BatchNorm): also the
    eval-mode output AFTER this one training-mode forward (running statistics updated once) and the buffers."""
    model.zero_grad()
    x = x.clone().requires_grad_(True)
    y = model(x)
    out = {"x": x.detach().numpy(), "y": y.detach().numpy()}
    if with_grads:
        t = formula_tensor(tuple(y.shape), 77, 1.0)
        (y * t).sum().backward()
        out["dx"] = x.grad.numpy()
        names, sums = [], []
        for name, p in model.named_parameters():
            if p.grad is None:
                continue
            names.append(name)
            sums.append(grad_summary(p.grad))
            if full_grads:
                out["g:" + name] = p.grad.numpy()
        out["grad_names"] = np.array(names)
        out["grad_sums"] = np.array(sums, dtype=np.float64)
    if with_eval:
        model.eval()
        with torch.no_grad():
            out["y_eval"] = model(x.detach()).numpy()
        model.train()
        bnames = [n for n, _ in model.named_buffers()]
        out["buffer_names"] = np.array(bnames)
        out["buffer_sums"] = np.array([grad_summary(b.double()) for _, b in model.named_buffers()], dtype=np.float64)
    return out


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  {name}.npz  {os.path.getsize(path)/1024:.1f} KiB")


def main():
    torch.set_num_threads(8)
    models = import_reference()
    from models import common, rcan, rdn, wdsr  # reference modules

    manifest = {}
    # GOLDEN_ONLY=<substring>[,<substring>...]: regenerate only the model cases whose name contains one of them and merge
    # their entries into the existing manifest (blocks / trajectories are left alone)
    only = [t for t in os.environ.get("GOLDEN_ONLY", "").split(",") if t]
    if only:
        with open(os.path.join(OUT, "manifest.json")) as f:
            manifest = json.load(f)

    # ---- (1) blocks at true width, small spatial extent -------------------
    print("blocks:")
    blocks = {
        "block_resblock64_rs0.1": (common.ResBlock(n_feats=64, res_scale=0.1), (2, 64, 12, 12)),
        "block_resblock256": (common.ResBlock(n_feats=256, res_scale=1.0), (1, 256, 8, 8)),
        "block_upscale4_64": (common.UpscaleBlock(4, 64), (2, 64, 6, 6)),
        "block_upscale3_64": (common.UpscaleBlock(3, 64), (1, 64, 5, 7)),
        "block_upscale2_64": (common.UpscaleBlock(2, 64), (1, 64, 6, 6)),
        "block_calayer64_r16": (rcan.CALayer(64, 16), (2, 64, 10, 10)),
        "block_rcab64": (rcan.RCAB(common.DefaultConv2d, 64, 3, 16), (2, 64, 10, 10)),
        "block_resgroup64_b2": (rcan.ResidualGroup(common.DefaultConv2d, 64, 3, 16, act=None,
                                                   res_scale=1, n_resblocks=2), (1, 64, 9, 9)),
        "block_rdb_64_64_8": (rdn._RDB(64, 64, 8), (1, 64, 8, 8)),
        "block_rdb_16_8_3": (rdn._RDB(16, 8, 3), (2, 16, 7, 9)),
        "block_wdsr_a128": (wdsr._Block_A(128, 3, wn=nn.utils.weight_norm, res_scale=1), (1, 128, 8, 8)),
        "block_wdsr_b128": (wdsr._Block_B(128, 3, wn=nn.utils.weight_norm, res_scale=1), (1, 128, 8, 8)),
    }
    for name, (mod, shp) in ([] if only else blocks.items()):
        formula_fill(mod)
        # feature maps: centred values, like post-conv activations
        x = formula_tensor(shp, 1000, 1.0)
        save(name, **run_case(mod, x))
    ms = common.MeanShift()
    ma = common.MeanShift(sign=1)
    x = formula_input((2, 3, 6, 6))
    if not only:
        save("block_meanshift", x=x.numpy(), y_sub=ms(x).detach().numpy(), y_add=ma(x).detach().numpy(),
           sub_weight=ms.weight.detach().numpy(), sub_bias=ms.bias.detach().numpy(),
           add_weight=ma.weight.detach().numpy(), add_bias=ma.bias.detach().numpy())

    # ---- (2) full models: reduced and full-size, formula-filled -----------
    print("models:")
    cases = {
        # reduced (ctor-expressible) variants, all scales the class supports
        "edsr_f16_b2_x2": ("EDSR", dict(n_feats=16, n_resblocks=2, res_scale=0.1, scale_factor=2), (2, 3, 10, 12)),
        "edsr_f16_b2_x3": ("EDSR", dict(n_feats=16, n_resblocks=2, res_scale=0.1, scale_factor=3), (1, 3, 9, 8)),
        "edsr_f16_b2_x4": ("EDSR", dict(n_feats=16, n_resblocks=2, res_scale=0.1, scale_factor=4), (2, 3, 8, 8)),
        "edsr_f16_b2_x8": ("EDSR", dict(n_feats=16, n_resblocks=2, res_scale=1, scale_factor=8), (1, 3, 6, 6)),
        "edsr_f16_b2_x4_gray": ("EDSR", dict(n_feats=16, n_resblocks=2, res_scale=1, scale_factor=4, channels=1), (1, 1, 8, 8)),
        "rcan_f16_g2_b2_r4_x4": ("RCAN", dict(n_feats=16, n_resblocks=2, n_resgroups=2, reduction=4, scale_factor=4), (2, 3, 8, 8)),
        "rcan_f16_g2_b2_r4_x2": ("RCAN", dict(n_feats=16, n_resblocks=2, n_resgroups=2, reduction=4, scale_factor=2), (1, 3, 9, 11)),
        "wdsr_b_f16_b2_x4": ("WDSR", dict(type="B", n_feats=16, n_resblocks=2, scale_factor=4), (2, 3, 8, 8)),
        "wdsr_a_f16_b2_x4": ("WDSR", dict(type="A", n_feats=16, n_resblocks=2, scale_factor=4), (2, 3, 8, 8)),
        "wdsr_b_f16_b2_x2": ("WDSR", dict(type="B", n_feats=16, n_resblocks=2, scale_factor=2), (1, 3, 9, 7)),
        "wdsr_b_f16_b2_x3": ("WDSR", dict(type="B", n_feats=16, n_resblocks=2, scale_factor=3), (1, 3, 6, 6)),
        "rdn_a_g0_16_x4": ("RDN", dict(rdn_config="A", G0=16, scale_factor=4), (1, 3, 8, 8)),
        "rdn_a_g0_16_x2": ("RDN", dict(rdn_config="A", G0=16, scale_factor=2), (1, 3, 7, 8)),
        "rdn_a_g0_16_x3": ("RDN", dict(rdn_config="A", G0=16, scale_factor=3), (1, 3, 6, 6)),
        "srcnn_x2": ("SRCNN", dict(scale_factor=2), (4, 3, 32, 32)),
        "srcnn_x4": ("SRCNN", dict(scale_factor=4), (1, 3, 12, 12)),
        # full-size BASELINE configs (run_comparisons.sh:34-45)
        "edsr_baseline_x4": ("EDSR", dict(n_feats=64, n_resblocks=16, res_scale=0.1, scale_factor=4), (1, 3, 12, 12)),
        "edsr_large_x4": ("EDSR", dict(n_feats=256, n_resblocks=32, res_scale=0.1, scale_factor=4), (1, 3, 8, 8)),
        "rcan_full_x4": ("RCAN", dict(n_feats=64, reduction=16, n_resgroups=10, n_resblocks=20, scale_factor=4), (1, 3, 8, 8)),
        "wdsr_b_full_x4": ("WDSR", dict(type="B", scale_factor=4), (1, 3, 12, 12)),
        "wdsr_a_full_x4": ("WDSR", dict(type="A", scale_factor=4), (1, 3, 8, 8)),
        "rdn_b_full_x4": ("RDN", dict(rdn_config="B", scale_factor=4), (1, 3, 8, 8)),
        "rdn_a_full_x4": ("RDN", dict(rdn_config="A", scale_factor=4), (1, 3, 8, 8)),
        # SRResNet (BatchNorm + PReLU, 9x9 convs; training-mode forward/backward + eval-mode output)
        # and D-DBPN (strided / transposed projection convs).  DDBPN has no width/depth ctor arguments.
        "srresnet_f16_b2_x4": ("SRResNet", dict(n_feats=16, n_resblocks=2, scale_factor=4), (2, 3, 8, 8)),
        "srresnet_f16_b2_x2": ("SRResNet", dict(n_feats=16, n_resblocks=2, scale_factor=2), (1, 3, 9, 7)),
        "srresnet_f16_b2_x3": ("SRResNet", dict(n_feats=16, n_resblocks=2, scale_factor=3), (2, 3, 6, 6)),
        "srresnet_full_x4": ("SRResNet", dict(scale_factor=4), (2, 3, 12, 12)),
        "ddbpn_x2": ("DDBPN", dict(scale_factor=2), (1, 3, 6, 7)),
        "ddbpn_x4": ("DDBPN", dict(scale_factor=4), (1, 3, 6, 6)),
        "ddbpn_x8": ("DDBPN", dict(scale_factor=8), (1, 3, 4, 4)),
    }
    for name, (cls, kw, shp) in cases.items():
        if only and not any(t in name for t in only):
            continue
        torch.manual_seed(0)
        m = getattr(models, cls)(**kw)
        # init checksums BEFORE the formula fill: pins torch-default init under seed 0
        init = {n: [float(p.double().sum()), float(p.double().abs().sum())] + [float(v) for v in p.flatten()[:3]]
                for n, p in m.state_dict().items()}
        manifest[name] = {
            "class": cls, "kwargs": kw, "input_shape": list(shp),
            "state_dict": [[n, list(p.shape)] for n, p in m.state_dict().items()],
            "trainable": [n for n, p in m.named_parameters() if p.requires_grad],
            "n_params_trainable": sum(p.numel() for p in m.parameters() if p.requires_grad),
            "init_seed0": init if len(init) < 400 else {k: init[k] for k in list(init)[:40]},
        }
        formula_fill(m)
        x = formula_input(shp)
        small = manifest[name]["n_params_trainable"] < 60000
        save("model_" + name, **run_case(m, x, with_grads=True, full_grads=small, with_eval=bool(list(m.named_buffers()))))

    # ---- (3) three-step training trajectories via the reference's own hooks
    print("trajectories:")
    for tag, losses, opt in ([] if only else (("l1_adam", "l1", "ADAM"), ("l2_sgd", "0.5*l2 + 0.5*l1", "SGD"))):
        m = models.EDSR(n_feats=16, n_resblocks=2, res_scale=0.1, scale_factor=4, losses=losses,
                        optimizer=opt, optimizer_params=["lr=1e-2"] if opt == "SGD" else ["lr=1e-4"])
        formula_fill(m)
        kw = {}
        if opt == "SGD":
            # the reference drops optimizer_params (srmodel.py:602-603) and SGD has no default lr
            # in older torch; in torch 2.10 the default is 1e-3.
            pass
        optim = m.configure_optimizers()[0]
        rec = {"opt_defaults": json.dumps({k: (list(v) if isinstance(v, tuple) else v)
                                           for k, v in optim.defaults.items()
                                           if isinstance(v, (int, float, tuple, bool, type(None)))})}
        losses_out = []
        for step in range(3):
            lr_ = formula_input((2, 3, 8, 8), k=2000 + step)
            hr_ = formula_input((2, 3, 32, 32), k=3000 + step)
            optim.zero_grad()
            res = m.training_step({"lr": lr_, "hr": hr_, "path": ["a", "b"]}, step)
            losses_out.append([float(res["loss"])] + [float(v) for k2, v in sorted(res.items()) if k2 != "loss"])
            rec[f"keys{step}"] = np.array(sorted(res.keys()))
            res["loss"].backward()
            optim.step()
        rec["losses"] = np.array(losses_out, dtype=np.float64)
        for n, p in m.state_dict().items():
            rec["w:" + n] = p.detach().numpy()
        save("traj_edsr_f16_b2_x4_" + tag, **rec)

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("manifest.json written;", len(manifest), "model cases")


if __name__ == "__main__":
    main()
