#!/usr/bin/env python3
"""Train an SR model on the MI355X hot path (flag vocabulary of the reference's train.py:289-303).

Data: the reference's data module (PIL / HuggingFace datasets, srdata.py) is out of scope (SURVEY.md section 2
row 11) and there is no network here, so `--synthetic` (default) draws uniform patches of the BASELINE shape.
Launch 1 process per GPU with torchrun for data-parallel training (RCCL over xGMI).
"""
import argparse
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    import sr_amd
    from sr_amd import trainer as T
    p = argparse.ArgumentParser()
    p.add_argument("-m", "--model", default="EDSR", choices=[m for m in sr_amd.models.__all__ if m != "SRModel"])
    p.add_argument("-s", "--scale_factor", type=int, default=4)
    p.add_argument("--patch_size", type=int, default=192, help="HR patch edge (LR = patch_size // scale_factor)")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--precision", default="bf16")
    p.add_argument("--losses", default="l1")
    p.add_argument("--optimizer", default="ADAM")
    p.add_argument("--max_steps", type=int, default=100)
    p.add_argument("--checkpoint", default=None, help="state_dict (.pt / Lightning .ckpt) to start from")
    p.add_argument("--save", default=None)
    p.add_argument("--n_feats", type=int, default=None)
    p.add_argument("--n_resblocks", type=int, default=None)
    p.add_argument("--res_scale", type=float, default=None)
    a = p.parse_args()
    kw = {k: getattr(a, k) for k in ("n_feats", "n_resblocks", "res_scale") if getattr(a, k) is not None}
    model = getattr(sr_amd, a.model)(scale_factor=a.scale_factor, patch_size=a.patch_size, batch_size=a.batch_size,
                                     precision=a.precision, losses=a.losses, optimizer=a.optimizer, **kw)
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=True)
    tr = T.Trainer(max_steps=a.max_steps, log_every=10)
    lr = a.patch_size // a.scale_factor
    batches = (T.synthetic_batch(a.batch_size, 3, lr, a.scale_factor, 1234 + 7919 * s + tr.rank, "cpu") for s in range(a.max_steps))
    tr.fit(model, batches)
    if a.save and tr.rank == 0:
        torch.save({"state_dict": model.state_dict()}, a.save)


if __name__ == "__main__":
    main()
