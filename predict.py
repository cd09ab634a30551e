#!/usr/bin/env python3
"""Super-resolve images with a trained model (flag vocabulary of the reference's predict.py:172-190).

    python predict.py -m edsr --checkpoint edsr.ckpt --predict_datasets /data/Set5_LR /data/Set14_LR

Every `--predict_datasets` entry is a directory of LR images (PNG/JPEG) or of `.npy` CHW float arrays; each image goes
through `SRModel.predict_step` (forward, clamp, srmodel.py:375-433), which writes
`<default_root_dir>/<dataset name>/<image name>.png` (and the 96x96 centre crop `_center.png`) with
torchvision.utils.save_image's rounding, like the reference.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    import numpy as np
    import torch
    import sr_amd
    from sr_amd import data as D
    p = argparse.ArgumentParser()
    p.add_argument("-m", "--model", default="srcnn")
    p.add_argument("-s", "--scale_factor", type=int, default=4)
    p.add_argument("--checkpoint", default="", help="state_dict (.pt / Lightning .ckpt)")
    p.add_argument("--precision", default="bf16")
    p.add_argument("--accelerator", default="auto", choices=("auto", "gpu", "cpu"))
    p.add_argument("--default_root_dir", default="results")
    p.add_argument("--predict_datasets", nargs="+", required=True, help="directories of LR images")
    p.add_argument("--n_feats", type=int, default=None)
    p.add_argument("--n_resblocks", type=int, default=None)
    p.add_argument("--n_resgroups", type=int, default=None)
    p.add_argument("--res_scale", type=float, default=None)
    a = p.parse_args(argv)
    kw = {k: getattr(a, k) for k in ("n_feats", "n_resblocks", "n_resgroups", "res_scale") if getattr(a, k) is not None}
    names = {m.lower(): m for m in sr_amd.models.__all__ if m != "SRModel"}
    cls = getattr(sr_amd, names[a.model.lower()])
    ds_names = [os.path.basename(os.path.normpath(d)) for d in a.predict_datasets]
    model = cls(scale_factor=a.scale_factor, precision=a.precision, default_root_dir=a.default_root_dir, predict_datasets=ds_names, **kw)
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=True)
    use_gpu = a.accelerator == "gpu" or (a.accelerator == "auto" and torch.cuda.is_available() and cls is not sr_amd.SRCNN)
    dev = torch.device("cuda" if use_gpu else "cpu")
    model = model.to(dev).eval()
    from PIL import Image
    n = 0
    for di, d in enumerate(a.predict_datasets):
        for f in sorted(os.listdir(d)):
            stem, ext = os.path.splitext(f)
            if ext.lower() == ".npy":
                x = torch.from_numpy(np.load(os.path.join(d, f))).float()
            elif ext.lower() in (".png", ".jpg", ".jpeg", ".bmp"):
                x = D.image_to_tensor(np.asarray(Image.open(os.path.join(d, f)).convert("RGB")))
            else:
                continue
            with torch.no_grad():
                model.predict_step({"lr": x[None].to(dev), "path": [stem]}, n, di)
            n += 1
    print(f"wrote {n} images under {a.default_root_dir}", flush=True)


if __name__ == "__main__":
    main()
