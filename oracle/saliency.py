"""CPU restatement of the reference's saliency-derived weights (`get_weights`,
src/patolette/patolette.pyx:203-313) -- SURVEY.md 8(f)-1.

TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/): imported by tests/ and by
__graft_entry__.smoke(), never by the product package.

PARITY UNPINNED: the reference's binding cannot be imported in this image (`skimage` is absent,
patolette.pyx:4-5), so no output of the reference itself pins this file.  It follows the
reference statement by statement and calls the *same* third-party routines wherever they exist
here (numpy `mean` / `cov` / `linalg.inv`, scipy `cdist(..., 'mahalanobis')`); only
`skimage.color.rgb2lab` (published algorithm: sRGB companding -> XYZ (D65, 2 deg) -> CIELAB) and
the Cython raster scans (oracle/patolette_oracle.c `orc_mbd`) are restated.
"""
from math import exp, floor, sqrt

import numpy as np
from scipy.spatial.distance import cdist

from . import binding

# skimage.color.colorconv: xyz_from_rgb (sRGB primaries, D65) and the D65 / 2-degree white point
_XYZ_FROM_RGB = np.array([[0.412453, 0.357580, 0.180423],
                          [0.212671, 0.715160, 0.072169],
                          [0.019334, 0.119193, 0.950227]])
_WHITE_D65_2 = np.array([0.95047, 1.0, 1.08883])


def rgb2lab(rgb):
    """skimage.color.rgb2lab for float64 sRGB in [0,1] (rgb2xyz then xyz2lab, illuminant D65, observer 2)."""
    arr = np.array(rgb, dtype=np.float64, copy=True)
    mask = arr > 0.04045
    arr[mask] = np.power((arr[mask] + 0.055) / 1.055, 2.4)
    arr[~mask] /= 12.92
    xyz = arr @ _XYZ_FROM_RGB.T
    xyz = xyz / _WHITE_D65_2
    mask = xyz > 0.008856
    xyz[mask] = np.cbrt(xyz[mask])
    xyz[~mask] = 7.787 * xyz[~mask] + 16.0 / 116.0
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    L = (116.0 * y) - 16.0
    a = 500.0 * (x - y)
    b = 200.0 * (y - z)
    return np.concatenate([v[..., np.newaxis] for v in (L, a, b)], axis=-1)


def _border_contrast(lab, lab_rows, region):
    """Mahalanobis distance of every pixel to one border region's mean colour (patolette.pyx:228-270)."""
    mean = np.mean(region, axis=(0, 1))
    flat = region.reshape((region.shape[0] * region.shape[1], 3))
    vi = np.linalg.inv(np.cov(flat.T))
    m2 = np.zeros((1, 3))
    m2[0, :] = mean
    u = cdist(lab_rows, m2, 'mahalanobis', VI=vi)
    return u.reshape((lab.shape[0], lab.shape[1]))


def check_shape(rows, cols):
    """Shapes the reference's get_weights cannot process (it raises there too, with other types):
    rows or cols <= 3 (mbd returns None, :157-158), a zero-thickness border (NaN statistics), or a
    border thicker than the image allows (the reshapes at :228-232 fail)."""
    bt = int(floor(0.1 * sqrt(rows * cols)))
    if rows <= 3 or cols <= 3:
        return "saliency weights need an image larger than 3 pixels in both dimensions"
    if bt < 1:
        return "saliency weights need at least 100 pixels (the border band would be empty)"
    if rows < bt + 1 or cols < bt + 1:
        return "saliency weights: image too elongated for its border band"
    return None


def get_weights(img, tile_size):
    """img: (rows, cols, 3) float64 sRGB in [0,1].  Returns rows*cols float64 weights (patolette.pyx:203-313)."""
    img = np.asarray(img, dtype=np.float64)
    rows, cols = img.shape[0], img.shape[1]
    err = check_shape(rows, cols)
    if err:
        raise ValueError(err)
    img_mean = np.mean(img, axis=2).astype(np.float32)                       # :204
    sal = binding.mbd(img_mean, 3)                                           # :205

    img_size = sqrt(rows * cols)                                             # :210
    bt = int(floor(0.1 * img_size))                                          # :211
    lab = rgb2lab(img)                                                       # :213
    # the four bands, named as the reference names them (:215-219): "left" is the top rows, "right" the
    # rows ending one short of the last, "top" the first columns, "bottom" the columns ending one short
    px_left = lab[0:bt, :, :]
    px_right = lab[rows - bt - 1:-1, :, :]
    px_top = lab[:, 0:bt, :]
    px_bottom = lab[:, cols - bt - 1:-1, :]
    lab_rows = lab.reshape(rows * cols, 3)
    u_left = _border_contrast(lab, lab_rows, px_left)
    u_right = _border_contrast(lab, lab_rows, px_right)
    u_top = _border_contrast(lab, lab_rows, px_top)
    u_bottom = _border_contrast(lab, lab_rows, px_bottom)

    f32 = lambda v: float(np.float32(v))            # `cdef float` locals (:272-275, :288-289)   # noqa: E731
    u_left = u_left / f32(np.max(u_left))
    u_right = u_right / f32(np.max(u_right))
    u_top = u_top / f32(np.max(u_top))
    u_bottom = u_bottom / f32(np.max(u_bottom))
    u_max = np.maximum(np.maximum(np.maximum(u_left, u_right), u_top), u_bottom)
    u_final = (u_left + u_right + u_top + u_bottom) - u_max                  # :284-286
    u_max_final = f32(np.max(u_final))
    sal_max = f32(np.max(sal))
    # float32 array / Python float stays float32 (:291), the sum with the float64 map widens
    s = (sal / np.float32(sal_max)).astype(np.float32).astype(np.float64) + u_final / u_max_final
    s = s / np.max(s)                                                        # :292
    xv, yv = np.meshgrid(np.arange(cols), np.arange(rows))                   # :294
    w2 = rows / 2.0
    h2 = cols / 2.0
    c = 1 - np.sqrt(np.power(xv - h2, 2) + np.power(yv - w2, 2)) / sqrt(np.power(w2, 2) + np.power(h2, 2))   # :300
    s = s * c
    s = s / np.max(s)                                                        # :309
    s = np.array([1.0 / (1.0 + exp(-10.0 * (v - 0.5))) for v in s.reshape(-1)])   # :304-310 (math.exp per element)
    return 1 + s ** 2 * (rows * cols) / tile_size ** 2                       # :313
