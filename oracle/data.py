"""CPU restatement of the training data step that feeds the path (oracle: test infrastructure).

Reference: `_SRDataset._get_item` / `_get_patch` (srdata.py:57-169), mode 'train' with augmentation:
  crop   lr_patch = TF.crop(lr, top=lr_x, left=lr_y, p, p);  hr_patch = TF.crop(hr, s*lr_x, s*lr_y, s*p, s*p)   (:137-169)
  rotate angle in {0,90,180,270}: TF.rotate(img, angle)  -- counter-clockwise, square patch            (:78-82)
  flips  TF.hflip, TF.vflip                                                                          (:84-91)
  to_tensor  uint8 HWC -> float32 CHW / 255                                                          (:124-128)
torchvision is not installable here; its functional ops on PIL images are PIL calls (Image.crop / rotate /
transpose), so this restatement is pinned against PIL itself in tests/test_oracle_data.py.  The random draws
(random.randrange / random.choice) are outside the arithmetic: parameters are explicit arguments here.
"""
import numpy as np


def get_patch_pair(lr, hr, top, left, patch_lr, scale, angle=0, hflip=False, vflip=False):
    """lr, hr: uint8 arrays [H, W, C].  Returns float32 CHW arrays in [0, 1]."""
    p, s = patch_lr, scale
    lp = lr[top:top + p, left:left + p]
    hp = hr[s * top:s * top + s * p, s * left:s * left + s * p]
    assert lp.shape[:2] == (p, p) and hp.shape[:2] == (s * p, s * p), "patch outside the image"
    k = (angle // 90) % 4
    if k:
        lp, hp = np.rot90(lp, k), np.rot90(hp, k)          # counter-clockwise, like PIL.Image.rotate
    if hflip:
        lp, hp = lp[:, ::-1], hp[:, ::-1]
    if vflip:
        lp, hp = lp[::-1], hp[::-1]
    to_t = lambda a: np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32) / 255.0
    return to_t(lp), to_t(hp)
