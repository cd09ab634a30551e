"""Pins oracle/data.py against PIL, the library torchvision's functional crop / rotate / hflip / vflip call on PIL
images (srdata.py:78-91,166-167), for every angle x flip combination."""
import itertools

import numpy as np
import pytest
from PIL import Image

from oracle import data as OD


@pytest.mark.parametrize("angle,hf,vf", list(itertools.product((0, 90, 180, 270), (False, True), (False, True))))
def test_patch_pair_matches_pil(angle, hf, vf):
    rng = np.random.default_rng(3)
    s, p = 4, 6
    lr = rng.integers(0, 256, (20, 31, 3), dtype=np.uint8)
    hr = rng.integers(0, 256, (80, 124, 3), dtype=np.uint8)
    top, left = 5, 17

    def pil_path(img, t, l, size):
        im = Image.fromarray(img).crop((l, t, l + size, t + size))        # TF.crop(img, top, left, h, w)
        if angle:
            im = im.rotate(angle)                                         # TF.rotate: counter-clockwise, nearest
        if hf:
            im = im.transpose(Image.FLIP_LEFT_RIGHT)                      # TF.hflip
        if vf:
            im = im.transpose(Image.FLIP_TOP_BOTTOM)                      # TF.vflip
        return np.asarray(im).transpose(2, 0, 1).astype(np.float32) / 255.0   # TF.to_tensor
    lo, ho = OD.get_patch_pair(lr, hr, top, left, p, s, angle, hf, vf)
    np.testing.assert_array_equal(lo, pil_path(lr, top, left, p))
    np.testing.assert_array_equal(ho, pil_path(hr, s * top, s * left, s * p))
