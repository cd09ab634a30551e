"""The data step in front of the path, on the GPU.

Reference: `_SRDataset._get_item` / `_get_patch` in 'train' mode (srdata.py:64-91,137-169): random LR/HR patch
pair, rot90 / hflip / vflip augmentation, `to_tensor`.  The reference runs this in `cpu_count()//2` PIL worker
processes (srdata.py:514-516); at >20k patches/s per GPU those cannot keep up, so here the images live on the
GPU as uint8 HWC tensors and ONE kernel launch (`srk_sample_patches`) produces the whole NCHW fp32 batch.

Random draws follow the reference's order per sample (`random.randrange` for the patch origin, `random.choice`
for angle / hflip / vflip), with the origin drawn as (row, column) from (height, width) -- the reference's ndarray /
tensor branch (srdata.py:151-154); its PIL branch swaps the two (`lr_h, lr_w = lr_image.size`, :149-150).
"""
import ctypes as C
import math
import os
import random

import torch

from . import _lib as L


def shard_indices(n, rank=0, world=1, *, shuffle=True, seed=0, epoch=0, drop_last=False):
    """The sample indices rank `rank` of `world` processes visits in one epoch: the index split of
    torch.utils.data.DistributedSampler, which Lightning installs for the reference when `devices > 1`
    (configs/all.yml:127 `use_distributed_sampler: true`) -- same permutation (generator seeded with seed + epoch), same
    padding by wrap-around to a multiple of `world` (or truncation with drop_last), rank r takes positions r, r+world, ...
    Every rank sees the same number of samples, the union covers the data set."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    if drop_last and n % world != 0:
        per = math.ceil((n - world) / world)
    else:
        per = math.ceil(n / world)
    total = per * world
    if not drop_last:
        pad = total - len(idx)
        if pad > 0:
            idx += (idx * math.ceil(pad / max(len(idx), 1)))[:pad]
    else:
        idx = idx[:total]
    return idx[rank:total:world]


def load_image_pairs(directory, scale_factor, lr_directory=None):
    """PNG/JPEG files of `directory` as (lr, hr) uint8 HWC arrays.  HR is cropped to a multiple of the scale; LR comes
    from `lr_directory` (same file names) or is synthesised like the reference's on-the-fly path: PIL bicubic resize
    (srdata.py:222-231, TF.resize(..., BICUBIC) on a PIL image)."""
    import numpy as np
    from PIL import Image
    names = sorted(f for f in os.listdir(directory) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
    pairs, stems = [], []
    for f in names:
        hr = Image.open(os.path.join(directory, f)).convert("RGB")
        w, h = hr.size
        w, h = w - w % scale_factor, h - h % scale_factor
        hr = hr.crop((0, 0, w, h))
        if lr_directory is not None:
            lr = Image.open(os.path.join(lr_directory, f)).convert("RGB")
            assert lr.size == (w // scale_factor, h // scale_factor), f"Wrong sizes: LR {lr.size}, HR {hr.size}"
        else:
            lr = hr.resize((w // scale_factor, h // scale_factor), Image.BICUBIC)
        pairs.append((np.asarray(lr).copy(), np.asarray(hr).copy()))
        stems.append(os.path.splitext(f)[0])
    return pairs, stems


def image_to_tensor(arr):
    """uint8 HWC -> float CHW in [0,1] (torchvision's to_tensor, srdata.py:99-100)."""
    import numpy as np
    return torch.from_numpy(np.array(arr, copy=True)).permute(2, 0, 1).float() / 255.0


class PatchSampler:
    def __init__(self, pairs, scale_factor, patch_size, augment=True, device="cuda"):
        """pairs: list of (lr, hr) uint8 arrays/tensors [H, W, C]; patch_size = HR patch edge (srdata.py:149)."""
        assert patch_size % scale_factor == 0, f"patch_size ({patch_size}) should be divisible by scale_factor ({scale_factor})"
        self.scale, self.p = scale_factor, patch_size // scale_factor
        self.augment = augment
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("PatchSampler runs on the GPU (srk_sample_patches); there is no CPU fallback")
        self.pairs = []
        for lr, hr in pairs:
            lr = torch.as_tensor(lr, dtype=torch.uint8).contiguous().to(self.device)
            hr = torch.as_tensor(hr, dtype=torch.uint8).contiguous().to(self.device)
            assert lr.dim() == 3 and hr.dim() == 3 and lr.shape[2] == hr.shape[2]
            assert lr.shape[0] == hr.shape[0] // scale_factor and lr.shape[1] == hr.shape[1] // scale_factor, \
                f"Wrong sizes: LR {tuple(lr.shape[:2])}, HR {tuple(hr.shape[:2])}"
            assert lr.shape[0] >= self.p and lr.shape[1] >= self.p, "image smaller than the patch"
            self.pairs.append((lr, hr))
        self.channels = self.pairs[0][0].shape[2]

    def draw(self, index, rng=random):
        """The reference's random draws for one sample (srdata.py:157-158,78-91)."""
        lr, _ = self.pairs[index]
        top = rng.randrange(0, lr.shape[0] - self.p + 1)
        left = rng.randrange(0, lr.shape[1] - self.p + 1)
        angle, hf, vf = 0, False, False
        if self.augment:
            angle = rng.choice((0, 90, 180, 270))
            hf = rng.choice((True, False))
            vf = rng.choice((True, False))
        return top, left, angle, hf, vf

    def batch(self, indices, params=None, rng=random):
        """{'lr': [N,C,p,p], 'hr': [N,C,sp,sp]} fp32 on the GPU for the given image indices."""
        if params is None:
            params = [self.draw(i, rng) for i in indices]
        n = len(indices)
        descs = (L.PatchDesc * n)()
        for k, (i, (top, left, angle, hf, vf)) in enumerate(zip(indices, params)):
            lr, hr = self.pairs[i]
            descs[k] = L.PatchDesc(lr=lr.data_ptr(), hr=hr.data_ptr(), lr_h=lr.shape[0], lr_w=lr.shape[1],
                                   hr_h=hr.shape[0], hr_w=hr.shape[1], top=top, left=left, rot=(angle // 90) % 4,
                                   hflip=int(hf), vflip=int(vf))
        table = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(self.device)
        lr_out = torch.empty((n, self.channels, self.p, self.p), dtype=torch.float32, device=self.device)
        hr_out = torch.empty((n, self.channels, self.p * self.scale, self.p * self.scale), dtype=torch.float32, device=self.device)
        a = L.PatchArgs(table=table.data_ptr(), N=n, C=self.channels, patch_lr=self.p, scale=self.scale,
                        lr_out=lr_out.data_ptr(), hr_out=hr_out.data_ptr())
        L.call("srk_sample_patches", a, torch.cuda.current_stream().cuda_stream)
        return {"lr": lr_out, "hr": hr_out, "path": [f"pair{i}" for i in indices]}
