"""RNG-free tensors shared by the golden generator and the tests (oracle: test infrastructure).

Restates `formula_tensor` / `formula_fill` / `formula_input` of
tests/golden/generate_golden.py so that fixtures never need to store weights.
"""
import numpy as np
import torch


def formula_tensor(shape, k, amp):
    n = int(np.prod(shape)) if len(shape) else 1
    v = amp * np.sin(0.37 * np.arange(n, dtype=np.float64) + 0.61 * k)
    return torch.from_numpy(v.astype(np.float32)).reshape(tuple(shape))


def formula_input(shape, k=1000):
    return 0.5 + 0.5 * formula_tensor(shape, k, 1.0)


def fill_value(name, shape, k):
    """Value for the k-th entry of `named_parameters()` (k counts frozen ones too)."""
    if name.endswith("weight_g"):
        return 1.0 + 0.25 * formula_tensor(shape, k, 1.0)
    if len(shape) > 1:
        fan_in = int(np.prod(shape[1:]))
        return formula_tensor(shape, k, 1.0 / np.sqrt(fan_in))
    return formula_tensor(shape, k, 0.1)


def formula_fill_state_dict(sd, trainable):
    """In-place fill of an (ordered) state dict; `trainable` = names with requires_grad.

    Order and index k follow `module.named_parameters()`, which for every model
    on the path equals the state_dict order (no buffers on the hot path,
    SURVEY.md 8(a) a14).
    """
    order = getattr(sd, "param_order", None)
    if order is not None and (len(order) != len(sd) or any(a != b for a, b in zip(order, sd))):
        # BatchNorm buffers / modules registered twice (SRResNet): k is the index among named_parameters(), which counts
        # the frozen MeanShift tensors too -- they are not in `trainable`, so splice them in where the keys put them
        names = [n for n in sd if n in set(order) or (n.startswith(("sub_mean", "add_mean")))]
        seen, plist = set(), []
        for n in names:
            if id(sd[n]) in seen:
                continue
            seen.add(id(sd[n]))
            plist.append(n)
        for k, name in enumerate(plist):
            if name in trainable:
                with torch.no_grad():
                    sd[name].copy_(fill_value(name, tuple(sd[name].shape), k))
        return sd
    for k, (name, t) in enumerate(sd.items()):
        if name not in trainable:
            continue
        with torch.no_grad():
            t.copy_(fill_value(name, tuple(t.shape), k))
    return sd


def formula_fill_module(module):
    with torch.no_grad():
        for k, (name, p) in enumerate(module.named_parameters()):
            if p.requires_grad:
                p.copy_(fill_value(name, tuple(p.shape), k).to(p.device, p.dtype))
    return module
