#!/usr/bin/env python3
"""Train an SR model on the MI355X hot path (flag vocabulary of the reference's train.py:285-303 and of the Trainer
keys its configs use, configs/all.yml:83,125-127).

    python train.py -m edsr --devices 8 --batch_size 16 --max_steps 1000          # 8 ranks, RCCL over xGMI
    python train.py -m srcnn --accelerator cpu --max_steps 2                      # the CPU plumbing case

Data: `--train_dir` (a directory of HR images; LR is synthesised by PIL bicubic like srdata.py:222-231, patches are
drawn and augmented on the GPU by data.PatchSampler, the index space is split over ranks like DistributedSampler) or,
without it, synthetic uniform patches of the BASELINE shape (there is no network for DIV2K here).  `--val_dir`
runs validation_step over full images and prints the per-data-set metric means (on_validation_epoch_end).
`--devices N` with N > 1 starts N processes (one per GPU) through torch.distributed.run unless a launcher already did.
"""
import argparse
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-m", "--model", default="srcnn", help="model name, case-insensitive (models/__init__.py)")
    p.add_argument("-s", "--scale_factor", type=int, default=4)
    p.add_argument("--patch_size", type=int, default=128, help="HR patch edge (LR = patch_size // scale_factor)")
    p.add_argument("--batch_size", type=int, default=16)
    p.add_argument("--precision", default="bf16", help="32, 16 or bf16 (Trainer key of the reference)")
    p.add_argument("--losses", default="l1")
    p.add_argument("--optimizer", default="ADAM")
    p.add_argument("--max_steps", type=int, default=100)
    p.add_argument("--max_epochs", type=int, default=-1)
    p.add_argument("--devices", type=int, default=1, help="processes / GPUs on this node (Lightning Trainer key)")
    p.add_argument("--accelerator", default="auto", choices=("auto", "gpu", "cpu"))
    p.add_argument("--strategy", default="ddp", help="accepted for compatibility: data parallel is the only strategy")
    p.add_argument("--default_root_dir", default=".")
    p.add_argument("--checkpoint", default=None, help="state_dict (.pt / Lightning .ckpt) to start from")
    p.add_argument("--save", default=None, help="where rank 0 writes {'state_dict': ...} at the end")
    p.add_argument("--train_dir", default=None, help="directory of HR training images (default: synthetic patches)")
    p.add_argument("--val_dir", default=None, help="directory of HR validation images")
    p.add_argument("--eval_datasets", nargs="+", default=None)
    p.add_argument("--log_every", type=int, default=10)
    p.add_argument("--n_feats", type=int, default=None)
    p.add_argument("--n_resblocks", type=int, default=None)
    p.add_argument("--n_resgroups", type=int, default=None)
    p.add_argument("--res_scale", type=float, default=None)
    return p.parse_args(argv)


def spawn(a):
    """--devices N: N child processes (one per GPU) through torch.distributed.run; nothing has touched the GPU here."""
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.devices}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def main(argv=None):
    a = parse(argv)
    if a.devices > 1 and "WORLD_SIZE" not in os.environ:
        spawn(a)
    import torch
    import sr_amd
    from sr_amd import data as D
    from sr_amd import trainer as T
    names = {m.lower(): m for m in sr_amd.models.__all__ if m != "SRModel"}
    if a.model.lower() not in names:
        raise SystemExit(f"unknown model {a.model!r}; available: {', '.join(sorted(names))}")
    cls = getattr(sr_amd, names[a.model.lower()])
    use_gpu = a.accelerator == "gpu" or (a.accelerator == "auto" and torch.cuda.is_available() and cls is not sr_amd.SRCNN)
    kw = {k: getattr(a, k) for k in ("n_feats", "n_resblocks", "n_resgroups", "res_scale") if getattr(a, k) is not None}
    if a.eval_datasets:
        kw["eval_datasets"] = a.eval_datasets
    elif a.val_dir:
        kw["eval_datasets"] = [os.path.basename(os.path.normpath(a.val_dir))]
    model = cls(scale_factor=a.scale_factor, patch_size=a.patch_size, batch_size=a.batch_size, precision=a.precision,
                losses=a.losses, optimizer=a.optimizer, default_root_dir=a.default_root_dir, devices=a.devices, **kw)
    if a.checkpoint:
        sd = torch.load(a.checkpoint, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd), strict=True)
    tr = T.Trainer(device="cuda" if use_gpu else "cpu", max_steps=a.max_steps, log_every=a.log_every)
    lr_edge = a.patch_size // a.scale_factor

    if a.train_dir:
        pairs, _ = D.load_image_pairs(a.train_dir, a.scale_factor)
        if not use_gpu:
            raise SystemExit("--train_dir draws its patches on the GPU (data.PatchSampler); use synthetic data on CPU")
        sampler = D.PatchSampler(pairs, a.scale_factor, a.patch_size, augment=True, device=tr.device)

        def batches():
            step, epoch = 0, 0
            while a.max_steps < 0 or step < a.max_steps:
                mine = D.shard_indices(len(pairs), tr.rank, tr.world, shuffle=True, seed=0, epoch=epoch)
                for i in range(0, len(mine), a.batch_size):
                    yield sampler.batch(mine[i:i + a.batch_size])
                    step += 1
                    if 0 <= a.max_steps <= step:
                        return
                epoch += 1
                if 0 <= a.max_epochs <= epoch:
                    return
        data = batches()
    else:
        n = a.max_steps if a.max_steps >= 0 else 100
        data = (T.synthetic_batch(a.batch_size, 3, lr_edge, a.scale_factor, 1234 + 7919 * s + tr.rank, "cpu") for s in range(n))
    tr.fit(model, data)

    if a.val_dir:
        pairs, stems = D.load_image_pairs(a.val_dir, a.scale_factor)
        mine = D.shard_indices(len(pairs), tr.rank, tr.world, shuffle=False)
        model.eval()
        with torch.no_grad():
            for k, i in enumerate(mine):
                lr, hr = pairs[i]
                batch = {"lr": D.image_to_tensor(lr)[None].to(tr.device), "hr": D.image_to_tensor(hr)[None].to(tr.device), "path": [stems[i]]}
                model.validation_step(batch, k, 0)
        metrics = model.on_validation_epoch_end()
        if tr.world > 1:                 # every rank averaged its own shard of the images: combine, weighted by image count
            for k in sorted(metrics):
                t = torch.tensor([float(metrics[k]) * len(mine), float(len(mine))], dtype=torch.float64, device=tr.device)
                torch.distributed.all_reduce(t)
                metrics[k] = t[0] / t[1].clamp_min(1)
        if tr.rank == 0:
            print("validation: " + ", ".join(f"{k} {float(v):.4f}" for k, v in metrics.items()), flush=True)
    if a.save and tr.rank == 0:
        os.makedirs(os.path.dirname(os.path.abspath(a.save)), exist_ok=True)
        torch.save({"state_dict": model.state_dict()}, a.save)
    if tr.rank == 0 and tr.losses:
        print(f"done: {len(tr.losses)} steps, last loss {tr.losses[-1]:.6f}", flush=True)
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
