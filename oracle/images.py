"""TEST INFRASTRUCTURE (like everything under oracle/): smooth synthetic 'photographs' and the PSNR used to check north_star's
"PSNR within 0.01 dB of reference" on a TRAINED net (tests/test_gpu_fullsize_parity.py, bench.py's cpu_baseline leg).  The reference
measures PSNR on DIV2K / Set5 images (models/srmodel.py:224-232); there is no dataset in this image, so the images are sums of
low-frequency sin * cos products per channel plus a little noise, in [0, 1] -- smooth enough that a trained net super-resolves them."""
import torch


def smooth_images(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
    out = torch.zeros(n, 3, size, size)
    for i in range(n):
        for c in range(3):
            img = torch.zeros(size, size)
            for _ in range(6):
                fx, fy = (torch.rand(2, generator=g) * 9 + 0.5).tolist()
                px, py = (torch.rand(2, generator=g) * 6.28).tolist()
                amp = float(torch.rand(1, generator=g)) * 0.25
                img += amp * torch.sin(6.28 * fx * xx + px) * torch.cos(6.28 * fy * yy + py)
            out[i, c] = 0.5 + img
    out += 0.01 * torch.randn(out.shape, generator=g)
    return out.clamp(0, 1)


def psnr(a, b):
    """Mean over the images of 10 log10(1 / MSE), both clamped to [0, 1] (models/srmodel.py:224-232 without the luma / shave options)."""
    mse = ((a.double().clamp(0, 1) - b.double().clamp(0, 1)) ** 2).flatten(1).mean(1)
    return float((10.0 * torch.log10(1.0 / (mse + 1e-12))).mean())
