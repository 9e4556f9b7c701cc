"""TEST INFRASTRUCTURE — scalar CPU restatement of the two OpenCV calls either side of the hot path:

  resize_u8(img, (W,H))            cv2.resize(uint8, INTER_LINEAR)        datasets/SSHIDataset.py:19-22
  unit_float(img)                  `img[None]/255` then `.float()`        datasets/SSHIDataset.py:26-28, superpoint_glue_test.py:74
  warp_affine_u8(img, M, (W,H))    cv2.warpAffine(u8/255*255, M) + imwrite   superpoint_glue_test.py:100-113

OpenCV is third-party and absent from /root/reference (README.md:24-25 pins opencv-python 4.5.1.48) and from this
image: the functions restate its published fixed-point algorithms (resize.cpp 8U linear path: 11-bit weights,
`((b0*(S0>>4))>>16 + (b1*(S1>>4))>>16 + 2)>>2`; imgwarp.cpp warpAffine: AB_BITS 10, INTER_BITS 5, float tap table).
**Parity vs cv2 itself is unpinned**; the GPU kernels (csrc/ingest.hip) are held bit-exact to this restatement.
Pure-Python loops: small cases only."""
import math

import numpy as np


def _coef(d, scale, n, clamp_weight):
    f = np.float32((d + 0.5) * scale - 0.5)
    s = int(math.floor(f))
    f = np.float32(f - np.float32(s))
    if clamp_weight:
        if s < 0:
            f, s = np.float32(0), 0
        if s >= n - 1:
            f, s = np.float32(0), n - 1
    return s, int(np.rint((np.float32(1) - f) * np.float32(2048))), int(np.rint(f * np.float32(2048)))


def resize_u8(img, size_wh):
    W, H = size_wh
    Hs, Ws = img.shape
    sx, sy = 1.0 / (W / Ws), 1.0 / (H / Hs)
    out = np.zeros((H, W), np.uint8)
    for y in range(H):
        cy, b0, b1 = _coef(y, sy, Hs, False)
        y0, y1 = min(max(cy, 0), Hs - 1), min(max(cy + 1, 0), Hs - 1)
        for x in range(W):
            cx, a0, a1 = _coef(x, sx, Ws, True)
            x1 = min(cx + 1, Ws - 1)
            r0 = int(img[y0, cx]) * a0 + int(img[y0, x1]) * a1
            r1 = int(img[y1, cx]) * a0 + int(img[y1, x1]) * a1
            out[y, x] = min(max((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2, 0), 255)
    return out


def unit_float(img_u8):
    return (np.asarray(img_u8) / 255).astype(np.float32)


def _sat(v):
    return int(np.rint(min(max(v, -2147483648.0), 2147483647.0)))


def warp_affine_u8(img, M, size_wh):
    W, H = size_wh
    Hs, Ws = img.shape
    M = [float(v) for v in np.asarray(M, np.float64).ravel()]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    i0, i1, i3, i4 = M[4] * D, M[1] * (-D), M[3] * (-D), M[0] * D
    i2, i5 = -i0 * M[2] - i1 * M[5], -i3 * M[2] - i4 * M[5]
    src = np.asarray(img, np.float64) / 255 * 255
    out = np.zeros((H, W), np.uint8)
    for y in range(H):
        X0, Y0 = _sat((i1 * y + i2) * 1024.0) + 16, _sat((i4 * y + i5) * 1024.0) + 16
        for x in range(W):
            X, Y = (X0 + _sat(i0 * x * 1024.0)) >> 5, (Y0 + _sat(i3 * x * 1024.0)) >> 5
            ix, iy = X >> 5, Y >> 5
            fx, fy = np.float32(X & 31) / np.float32(32), np.float32(Y & 31) / np.float32(32)
            one = np.float32(1)
            acc = 0.0
            for dy, dx, w in ((0, 0, (one - fy) * (one - fx)), (0, 1, (one - fy) * fx), (1, 0, fy * (one - fx)), (1, 1, fy * fx)):
                yy, xx = iy + dy, ix + dx
                v = src[yy, xx] if 0 <= yy < Hs and 0 <= xx < Ws else 0.0
                acc = acc + v * float(w)
            out[y, x] = min(max(int(np.rint(acc)), 0), 255)
    return out
