"""`SRModel`: the reference's LightningModule base, restated for the MI355X build.

Mirrors models/srmodel.py:67-621 of the reference for the part of the surface the hot
path needs: constructor keywords, `forward` (abstract), `training_step`,
`validation_step`, `predict_step`, `configure_optimizers`, the loss-string parser and the
metric table.  Lightning is optional: when `lightning.pytorch` is importable the class
derives from `LightningModule` (so the reference's `main.py` / Trainer can drive it);
otherwise from `nn.Module` with no-op logging hooks and `trainer.py`'s own fit loop.

Out of scope: perceptual / adaptive losses, Comet /
TensorBoard image dumps, model-parallel flags (accepted and ignored with a warning).
"""
import itertools
import logging
from dataclasses import dataclass
from typing import Any, Callable

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim

from .. import optim as _hip_optim

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning.pytorch as pl
    _Base = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # noqa: BLE001
    HAVE_LIGHTNING = False

    class _Base(nn.Module):
        """Minimal stand-in for LightningModule (bookkeeping only)."""

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


@dataclass
class _SubLoss:
    name: str
    loss: Callable
    weight: float = 1.0


# models/srmodel.py:30-44 -- only the torch-native entries are on the path
def _l1_loss(sr, hr):
    """F.l1_loss; on the GPU the fused HIP forward/backward pair (ops.L1LossFn)."""
    if sr.is_cuda and sr.dtype == torch.float32 and sr.shape == hr.shape:
        from .. import ops
        return ops.l1_loss(sr, hr)
    return F.l1_loss(sr, hr)


_supported_losses = {"l1": _l1_loss, "l2": F.mse_loss, "mae": _l1_loss, "mse": F.mse_loss}
_out_of_scope_losses = {"adaptive", "dists", "edge_loss", "flip", "haarpsi", "lpips", "pencil_sketch", "pieapp"}

# models/srmodel.py:57-64
# 'ADAM' is torch.optim.Adam with the update of GPU parameters as one HIP launch (sr-pytorch-lightning_amd/optim.py)
_supported_optimizers = {"ADAM": _hip_optim.Adam, "RMSprop": optim.RMSprop, "SGD": optim.SGD}
_out_of_scope_optimizers = {"Ranger", "RangerVA", "RangerQH"}


def _psnr(x, y):
    """RGB PSNR, data_range 1, per image then batch mean (piq.psnr defaults; srmodel.py:52,582).
    On the GPU the squared-error reduction is the HIP kernel srk_image_sse (no D2H sync per image)."""
    if x.is_cuda:
        from .. import ops
        return ops.psnr(x, y)
    mse = ((x.float() - y.float()) ** 2).flatten(1).mean(dim=1)
    return (10.0 * torch.log10(1.0 / (mse + 1e-8))).mean()


def _ssim(x, y, kernel_size=11, sigma=1.5, k1=0.01, k2=0.03):
    """SSIM with piq.ssim's defaults (11x11 Gaussian, sigma 1.5, avg-pool by round(min(H,W)/256)).
    On the GPU: the HIP reduction srk_image_ssim."""
    if x.is_cuda and kernel_size == 11:
        from .. import ops
        return ops.ssim(x, y, sigma=sigma, k1=k1, k2=k2)
    x, y = x.float(), y.float()
    f = max(1, round(min(x.shape[-2:]) / 256))
    if f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    c = x.shape[1]
    co = torch.arange(kernel_size, dtype=torch.float32, device=x.device) - (kernel_size - 1) / 2.0
    g = torch.exp(-(co ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    k = torch.outer(g, g).view(1, 1, kernel_size, kernel_size).repeat(c, 1, 1, 1)
    c1, c2 = k1 ** 2, k2 ** 2
    mx, my = F.conv2d(x, k, groups=c), F.conv2d(y, k, groups=c)
    sxx = F.conv2d(x * x, k, groups=c) - mx ** 2
    syy = F.conv2d(y * y, k, groups=c) - my ** 2
    sxy = F.conv2d(x * y, k, groups=c) - mx * my
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    ss = (2 * mx * my + c1) / (mx ** 2 + my ** 2 + c1) * cs
    return ss.mean(dim=(-1, -2)).mean(dim=1).mean()


def _psnr_y(x, y, shave):
    """PSNR on BT.601 luma with a `shave`-pixel border removed (the SR community convention)."""
    def lum(t):
        return (65.481 * t[:, 0:1] + 128.553 * t[:, 1:2] + 24.966 * t[:, 2:3] + 16.0) / 255.0
    xl, yl = lum(x.float()), lum(y.float())
    if shave > 0:
        xl, yl = xl[..., shave:-shave, shave:-shave], yl[..., shave:-shave, shave:-shave]
    mse = ((xl - yl) ** 2).flatten(1).mean(dim=1)
    return (10.0 * torch.log10(1.0 / mse.clamp_min(1e-12))).mean()


def _psnr_y_metric(x, y):
    """PSNR-Y (BT.601 luma, border = 4 px): the quantity BASELINE.json names; NOT in the reference's table."""
    if x.is_cuda and x.shape[1] == 3:
        from .. import ops
        return ops.psnr(x, y, luma=True, shave=4, eps=0.0)
    return _psnr_y(x, y, 4)


_supported_metrics = {"PSNR": _psnr, "SSIM": _ssim, "PSNR-Y": _psnr_y_metric}   # srmodel.py:47-54 (+ PSNR-Y)


def _dtype_from_precision(precision):
    p = str(precision).lower()
    if p.startswith("bf16"):
        return torch.bfloat16
    if p.startswith("16"):
        return torch.float16
    if p.startswith("32"):
        return torch.float32
    raise ValueError(f"precision {precision!r} not understood (32, 16, 'bf16')")


class SRModel(_Base):
    """Base module for super-resolution models (reference: models/srmodel.py:67-143)."""

    def __init__(self,
                 batch_size: int = 16,
                 channels: int = 3,
                 default_root_dir: str = '.',
                 devices: None | list[int] | str | int = None,
                 eval_datasets: list[str] = ['DIV2K', 'Set5', 'Set14', 'B100', 'Urban100'],
                 log_loss_every_n_epochs: int = 5,
                 log_weights_every_n_epochs: int = 50,
                 losses: str = 'l1',
                 max_epochs: int = -1,
                 metrics: list[str] = ['PSNR', 'SSIM'],
                 metrics_for_pbar: list[str] = ['PSNR', 'SSIM'],
                 model_gpus: list[str] = [],
                 model_parallel: bool = False,
                 optimizer: str = 'ADAM',
                 optimizer_params: list[str] = [],
                 patch_size: int = 128,
                 precision: int | str = 32,
                 predict_datasets: list[str] = [],
                 save_results: int = -1,
                 save_results_from_epoch: str = 'last',
                 scale_factor: int = 4,
                 **kwargs: dict[str, Any]):
        super().__init__()
        self._logger = logging.getLogger(__name__)
        self.save_hyperparameters()
        # srmodel.py:105-108
        self.example_input_array = torch.zeros(batch_size, channels, patch_size // scale_factor, patch_size // scale_factor)
        if model_parallel:
            self._logger.warning("model_parallel is vestigial in the reference (SURVEY.md) and ignored here")
        self._model_parallel = False
        self._model_gpus = None
        self._batch_size = batch_size
        self._channels = channels
        self._default_root_dir = default_root_dir
        self._eval_datasets = eval_datasets
        self._last_epoch = max_epochs
        self._log_loss_every_n_epochs = log_loss_every_n_epochs
        self._log_weights_every_n_epochs = log_weights_every_n_epochs
        self._losses = self._create_losses(losses, patch_size, precision)
        self._metrics = self._create_metrics(metrics)
        self._metrics_for_pbar = metrics_for_pbar
        self._optim, self._optim_params = self._parse_optimizer_config(optimizer, optimizer_params)
        self._predict_datasets = predict_datasets
        self._save_results = save_results
        self._save_results_from_epoch = save_results_from_epoch
        self._scale_factor = scale_factor
        self._training_step_outputs = []
        self._validation_step_outputs = []
        #: arithmetic type of the HIP path: storage dtype of activations / packed weights (fp32 accumulate)
        self.compute_dtype = _dtype_from_precision(precision)
        #: storage dtype of the validation / predict forward.  bf16 keeps 8 mantissa bits on the residual trunk, which
        #: costs 0.005-0.013 dB of PSNR on a trained EDSR-baseline (tests/test_gpu_fullsize_parity.py); fp16 storage runs at the same
        #: speed and stays within 0.0002 dB of the fp32 reference path, so a bf16 model evaluates in fp16 unless told
        #: otherwise (`eval_precision=` 32 / 16 / 'bf16'; non-finite fp16 outputs fall back to the training dtype)
        ep = kwargs.get("eval_precision")
        self.eval_dtype = _dtype_from_precision(ep) if ep is not None else \
            (torch.float16 if self.compute_dtype == torch.bfloat16 else self.compute_dtype)

    # -- optimizers: srmodel.py:145-154 ------------------------------------------------------------
    def configure_optimizers(self):
        trainable = filter(lambda p: p.requires_grad, itertools.chain(self.parameters()))
        return [self._optim(trainable, **self._optim_params)]

    def forward(self, x):  # srmodel.py:156-158 (abstract)
        raise NotImplementedError

    def _pack_group(self):
        """Lazily created `ops.PackGroup` of the CURRENT storage dtype: one launch per step re-packs every conv's shadow
        weights.  One group per dtype, so the fp16 evaluation forward of a bf16 model (`_eval_forward`) never adds its
        entries to the group the (possibly hipGraph-captured) training step refreshes."""
        groups = self.__dict__.get("_srk_packs")
        if groups is None:
            groups = self.__dict__["_srk_packs"] = {}
        g = groups.get(self.compute_dtype)
        if g is None:
            from .. import ops
            g = groups[self.compute_dtype] = ops.PackGroup()
        return g

    # -- srmodel.py:160-171 ---------------------------------------------------------------------------
    def training_step(self, batch, batch_idx):
        img_sr = self.forward(batch['lr'])
        result = self._calculate_losses(img_sr=img_sr, img_hr=batch['hr'])
        return result

    def _eval_forward(self, x):
        """`forward` in the evaluation storage dtype (see `eval_dtype`)."""
        if self.eval_dtype == self.compute_dtype or not x.is_cuda:
            return self.forward(x)
        prev, self.compute_dtype = self.compute_dtype, self.eval_dtype
        try:
            y = self.forward(x)
        finally:
            self.compute_dtype = prev
        if self.eval_dtype == torch.float16 and not bool(torch.isfinite(y).all()):
            y = self.forward(x)                   # fp16 range exceeded: the training dtype's answer
        return y

    # -- srmodel.py:214-232 (metric core; image dumping is out of scope) ----------------------------
    def validation_step(self, batch, batch_idx, dataloader_idx=0):
        img_lr, img_hr = batch['lr'], batch['hr']
        img_sr = self._eval_forward(img_lr)
        assert img_sr.size() == img_hr.size(), \
            f'Output size for image {self._eval_datasets[dataloader_idx]}/{batch.get("path")} should be {img_hr.size()}, instead is {img_sr.size()}'
        img_hr = img_hr.clamp(0, 1)
        img_sr = img_sr.clamp(0, 1)
        result = self._calculate_metrics(img_sr=img_sr, img_hr=img_hr, dataloader_idx=dataloader_idx)
        self._validation_step_outputs.append(result)
        return result

    # -- srmodel.py:345-373 ---------------------------------------------------------------------------
    def on_validation_epoch_end(self):
        """Plain mean of the per-image metrics, per key (`<dataset>/<metric>`): what the reference logs at the end of a
        validation epoch.  The result is also kept in `last_validation_metrics` (no Lightning logger needed) and the
        step outputs are cleared (they hold device tensors)."""
        trainer = getattr(self, "_trainer", None)
        if not self._validation_step_outputs or (trainer is not None and getattr(trainer, "sanity_checking", False)):
            self._validation_step_outputs.clear()
            return {}

        def _mean(keys, metrics):
            out = {}
            for k in keys:
                vals = [m[k] for m in metrics if k in m]
                out[k] = torch.stack([torch.as_tensor(v).squeeze().float() for v in vals]).mean().cpu().detach()
            return out

        outs = self._validation_step_outputs
        metrics_dict = {}
        if isinstance(outs[0], dict):                       # one list of per-batch dicts (keys carry the dataset name)
            keys = []
            for m in outs:
                keys += [k for k in m if k not in keys]
            metrics_dict.update(_mean(keys, outs))
        else:                                               # list (per dataset) of lists of dicts
            for dataset_result in outs:
                metrics_dict.update(_mean(dataset_result[0].keys(), dataset_result))
        self.log_dict(metrics_dict, prog_bar=False, logger=True, add_dataloader_idx=False)
        self.last_validation_metrics = metrics_dict
        self._validation_step_outputs.clear()
        return metrics_dict

    # -- srmodel.py:375-433: forward + clamp + PNG on disk (the logger image dumps are out of scope) -------------
    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        img_sr = self._eval_forward(batch['lr']).clamp(0, 1)
        if self._predict_datasets and 'path' in batch and dataloader_idx < len(self._predict_datasets):
            from pathlib import Path
            local = Path(f'{self._default_root_dir}') / self._predict_datasets[dataloader_idx]
            local.mkdir(parents=True, exist_ok=True)
            name = batch['path'][0]
            self.save_png(img_sr[0], local / f'{name}.png')
            h, w = img_sr.shape[-2:]
            if h >= 96 and w >= 96:                          # K.CenterCrop(96) of the reference (srmodel.py:382-392)
                t, l = (h - 96) // 2, (w - 96) // 2
                self.save_png(img_sr[0, :, t:t + 96, l:l + 96], local / f'{name}_center.png')
        return img_sr

    @staticmethod
    def to_uint8(img):
        """torchvision.utils.save_image rounding (srmodel.py:311-315): floor(clamp(x,0,1)*255 + 0.5)."""
        return torch.floor(img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)

    @classmethod
    def save_png(cls, img, path):
        """One CHW float image -> PNG with torchvision.utils.save_image's rounding (srmodel.py:409-412), through PIL."""
        from PIL import Image
        u8 = cls.to_uint8(img.detach().float()).permute(1, 2, 0).cpu().numpy()
        Image.fromarray(u8[..., 0] if u8.shape[2] == 1 else u8).save(str(path))

    # the packed-weight group holds ctypes tables with device pointers: never copied / pickled with the module
    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_srk_packs", None)
        state.pop("_srk_wn", None)
        return state

    # -- srmodel.py:435-501 ------------------------------------------------------------------------------
    def _create_losses(self, losses_str: str, patch_size: int, precision=32) -> list[_SubLoss]:
        losses = []
        for loss in losses_str.split('+'):
            loss_split = loss.split('*')
            if len(loss_split) == 2:
                weight, loss_type = loss_split
                try:
                    weight = float(weight)
                except ValueError:
                    raise ValueError(f'{weight} is not a valid number to be used as weight for loss function {loss_type.strip()}')
            else:
                weight = 1.
                loss_type = loss_split[0]
            loss_type = loss_type.strip().lower()
            if loss_type in _supported_losses:
                fn = _supported_losses[loss_type]
            elif loss_type in _out_of_scope_losses:
                raise NotImplementedError(f'loss {loss_type} needs piq/kornia/robust_loss_pytorch and is outside this build '
                                          f'(SURVEY.md section 2 row 12). Supported: {", ".join(_supported_losses)}')
            else:
                raise AttributeError(f'Couldn\'t find loss {loss_type}. Supported losses: {", ".join(_supported_losses)}')
            losses.append(_SubLoss(name=loss_type, loss=fn, weight=weight))
        return losses

    def _create_metrics(self, metrics: list[str]):
        used = []
        for metric in metrics:
            if metric in _supported_metrics:
                used.append((metric, _supported_metrics[metric]))
            elif metric in {'BRISQUE', 'FLIP', 'LPIPS', 'MS-SSIM'}:
                raise NotImplementedError(f'metric {metric} needs piq and is outside this build. Supported: {", ".join(_supported_metrics)}')
            else:
                raise AttributeError(f'Couldn\'t find metric {metric}. Supported metrics: {", ".join(_supported_metrics)}')
        return used

    # -- srmodel.py:519-565 ------------------------------------------------------------------------------
    def _calculate_losses(self, img_sr: torch.Tensor, img_hr: torch.Tensor) -> dict[str, torch.Tensor]:
        # same values as the reference's `weight * loss` terms and their `sum()` (srmodel.py:519-565), without the launches a
        # multiplication by 1 and `0 + x` cost (two of the ~8 parameter-sized torch kernels of a batch-16 step)
        losses, names = [], []
        for l in self._losses:
            v = l.loss(img_sr, img_hr)
            losses.append(v if l.weight == 1 else l.weight * v)
            names.append(l.name)
        losses_dict = {f'loss/{k}': v for k, v in zip(names, losses)}
        total = losses[0]
        for v in losses[1:]:
            total = total + v
        losses_dict['loss'] = total
        return losses_dict

    # -- srmodel.py:567-593 ------------------------------------------------------------------------------
    def _calculate_metrics(self, img_sr, img_hr, dataloader_idx: int = 0):
        out = {}
        for name, metric in self._metrics:
            out[f'{self._eval_datasets[dataloader_idx]}/{name}'] = metric(img_sr, img_hr)
        return out

    # -- srmodel.py:595-621 ------------------------------------------------------------------------------
    def _parse_optimizer_config(self, optimizer: str, optimizer_params: list[str]):
        if optimizer in _supported_optimizers:
            optimizer_class = _supported_optimizers[optimizer]
        elif optimizer in _out_of_scope_optimizers:
            raise NotImplementedError(f'optimizer {optimizer} needs torch_optimizer and is outside this build')
        else:
            raise ValueError(f'Optimizer not recognized: {optimizer}. Supported optimizers: {", ".join(_supported_optimizers)}')
        # The reference re-binds `optimizer_params = {}` BEFORE iterating it (srmodel.py:602-603), so every
        # user-supplied entry is dropped and the optimizer runs at torch defaults.  Kept for parity
        # (pinned by tests/golden/traj_*): the argument is accepted and has no effect.
        return optimizer_class, {}
