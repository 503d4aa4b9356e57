"""CPU: `bricks.rotate_nearest` restates torchvision.transforms.functional.rotate (nearest, zero fill, no expand) as the
reference uses it to align prev_bev (modules/transformer.py:139-151); torchvision is not installed [3P].  SciPy's
`ndimage.rotate(order=0, reshape=False, mode='constant')` is an independent implementation of the same operation about the
same centre: direction of rotation, centre and fill are pinned by it (the two differ only in how a sampling position exactly
between two pixels is rounded: a few per cent of the pixels at non-trivial angles, none at 90 degrees)."""
import numpy as np
import pytest
import torch


@pytest.mark.parametrize("angle", [7.3, -12.0, 33.3, 90.0, 180.0, 1.0])
def test_rotate_nearest_agrees_with_scipy_in_direction_centre_and_fill(angle):
    from scipy import ndimage
    from vidar_amd.plugin.bricks import rotate_nearest
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.standard_normal((2, 40, 40)).astype(np.float32))
    got = rotate_nearest(img, angle, [20, 20]).numpy()
    same = np.stack([ndimage.rotate(img[c].numpy(), angle, reshape=False, order=0, mode="constant", cval=0.0) for c in range(2)])
    other = np.stack([ndimage.rotate(img[c].numpy(), -angle, reshape=False, order=0, mode="constant", cval=0.0) for c in range(2)])
    agree = float((got == same).mean())
    assert agree >= (0.999 if angle in (90.0, 180.0) else 0.93), agree
    if angle not in (180.0,) and abs(angle) > 5:
        assert float((got == other).mean()) < 0.5          # the opposite direction is a different image
    # zero fill outside the rotated square: the corners of a 33-degree rotation are empty in both
    if angle == 33.3:
        assert got[:, 0, 0].tolist() == [0.0, 0.0] and same[:, 0, 0].tolist() == [0.0, 0.0]
