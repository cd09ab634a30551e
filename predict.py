#!/usr/bin/env python3
"""Super-resolve tensors with a trained model (flag vocabulary of the reference's predict.py:175-188).

Inputs are `.pt`/`.npy` NCHW float images in [0,1] (PNG I/O needs torchvision/PIL pipelines that are out of
scope here); outputs are clamped, rounded to uint8 like torchvision.utils.save_image and saved as `.npy`.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import numpy as np
    import torch
    import sr_amd
    p = argparse.ArgumentParser()
    p.add_argument("-m", "--model", default="EDSR")
    p.add_argument("-s", "--scale_factor", type=int, default=4)
    p.add_argument("--checkpoint", required=True)
    p.add_argument("--precision", default="bf16")
    p.add_argument("--out_dir", default="results")
    p.add_argument("inputs", nargs="+")
    a = p.parse_args()
    model = getattr(sr_amd, a.model)(scale_factor=a.scale_factor, precision=a.precision)
    sd = torch.load(a.checkpoint, map_location="cpu")
    model.load_state_dict(sd.get("state_dict", sd), strict=True)
    model = model.cuda().eval()
    os.makedirs(a.out_dir, exist_ok=True)
    for path in a.inputs:
        x = torch.from_numpy(np.load(path)) if path.endswith(".npy") else torch.load(path)
        if x.dim() == 3:
            x = x[None]
        with torch.no_grad():
            sr = model.predict_step({"lr": x.float().cuda()}, 0)
        np.save(os.path.join(a.out_dir, os.path.splitext(os.path.basename(path))[0] + "_sr.npy"), model.to_uint8(sr).cpu().numpy())


if __name__ == "__main__":
    main()
