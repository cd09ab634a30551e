"""CPU restatement of the caller of the path: losses, training_step, optimizer (oracle: test infrastructure).

  parse_losses       `SRModel._create_losses` for the torch-only entries l1/l2/mae/mse
                     (models/srmodel.py:30-44,435-501): "w1*name1 + w2*name2", default weight 1
  calculate_losses   `SRModel._calculate_losses` (srmodel.py:519-565): {'loss/<name>': w*loss, 'loss': sum}
  make_optimizer     `SRModel.configure_optimizers` + `_parse_optimizer_config`
                     (srmodel.py:145-154,595-621).  The reference SHADOWS `optimizer_params` with
                     `{}` before parsing it (srmodel.py:602-603), so every user parameter is dropped and
                     the optimizer runs at torch defaults (Adam: lr 1e-3, betas (0.9,0.999), eps 1e-8).
                     Restated faithfully: `optimizer_params` is accepted and ignored.
  OracleModel        state dict + functional forward + training_step, used by tests and by
                     bench.py's cpu_baseline leg.
"""
import torch
import torch.nn.functional as F

from . import functional, init

_LOSSES = {"l1": F.l1_loss, "mae": F.l1_loss, "l2": F.mse_loss, "mse": F.mse_loss}
_OPTIMS = {"ADAM": torch.optim.Adam, "SGD": torch.optim.SGD, "RMSprop": torch.optim.RMSprop}


def parse_losses(losses_str):
    out = []
    for item in losses_str.split("+"):
        parts = item.split("*")
        if len(parts) == 2:
            try:
                weight = float(parts[0])
            except ValueError:
                raise ValueError(f"{parts[0]} is not a valid number to be used as weight for loss function {parts[1].strip()}")
            name = parts[1]
        else:
            weight, name = 1.0, parts[0]
        name = name.strip().lower()
        if name not in _LOSSES:
            raise AttributeError(f"Couldn't find loss {name}. Supported losses: {', '.join(_LOSSES)}")
        out.append((name, weight))
    return out


def calculate_losses(parsed, img_sr, img_hr):
    vals = {name: w * _LOSSES[name](img_sr, img_hr) for name, w in parsed}
    res = {f"loss/{k}": v for k, v in vals.items()}
    res["loss"] = sum(vals.values())
    return res


def make_optimizer(params, optimizer="ADAM", optimizer_params=()):
    if optimizer not in _OPTIMS:
        raise ValueError(f"Optimizer not recognized: {optimizer}. Supported optimizers: {', '.join(_OPTIMS)}")
    del optimizer_params            # dropped by the reference (srmodel.py:602-603)
    return _OPTIMS[optimizer]([p for p in params if p.requires_grad])


class OracleModel:
    """Reference-shaped model on CPU: `sd` (reference keys) + functional forward."""

    def __init__(self, cls, losses="l1", optimizer="ADAM", optimizer_params=(), **kw):
        self.cls, self.kw = cls, dict(kw)
        self.sd, self.trainable = init.build_state_dict(cls, **kw)
        for k in self.trainable:
            self.sd[k].requires_grad_(True)
        self._losses = parse_losses(losses)
        self._optimizer, self._optimizer_params = optimizer, optimizer_params

    def parameters(self):
        return [self.sd[k] for k in self.sd if k in self.trainable]

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k in self.sd:
                self.sd[k].copy_(sd[k])

    def forward(self, x):
        return functional.forward(self.cls, self.sd, x, **self.kw)

    __call__ = forward

    def training_step(self, batch, batch_idx=0):
        """models/srmodel.py:160-171"""
        return calculate_losses(self._losses, self.forward(batch["lr"]), batch["hr"])

    def configure_optimizers(self):
        return [make_optimizer(self.parameters(), self._optimizer, self._optimizer_params)]
