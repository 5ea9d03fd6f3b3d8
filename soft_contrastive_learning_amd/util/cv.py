"""Image geometry of the reference's loaders (util/cv.py:7-20) without OpenCV.

``load_images`` of the trainer (train/train.py:423-430) and of the feature extractor
(evaluation/inference.py:52-72) bring every image to the network's input size with

    resize_img(img, 240)            the longer side becomes 240 (NetVLAD head: any aspect ratio)
    standard_size(img, 180, 240)    cover 180 x 240, then centre crop (no NetVLAD head: fixed map)

both through ``cv2.resize(img, (0, 0), fx=scale, fy=scale)`` — INTER_LINEAR on uint8, i.e. plain
bilinear sampling at pixel centres WITHOUT any low-pass filter (a 1280 x 960 RobotCar frame comes
down by 5.33 x: the result aliases, and the released models were trained on exactly that).  PIL's
``resize(BILINEAR)`` widens its kernel by the scale factor, torch's ``interpolate(antialias=False)``
rounds differently: neither reproduces the reference's pixels, so the resampler is restated here.

``resize_linear`` follows OpenCV's generic 8-bit path (third-party, absent from /root/reference:
opencv-python is imported by util/cv.py:3 and util/io.py, no version pinned; algorithm as published
in modules/imgproc/src/resize.cpp — `resize`, `HResizeLinear`, `VResizeLinear<uchar, int, short,
FixedPtCast<int, uchar, 22>>`):
  * dsize = (round(W fx), round(H fy)), round half to even; the sampling scale is 1 / fx, NOT W / dsize;
  * source coordinate of destination d: float32((d + 0.5) / fx - 0.5), split into floor and fraction;
    left of the image the fraction is zeroed and the index clamped, same on the right (x); rows are
    clamped (y);
  * weights as 11-bit fixed point, short(round(w * 2048)); horizontal pass in int32;
  * vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
PARITY UNPINNED against OpenCV itself (not installed here); pinned by pencil cases and against a
float64 bilinear sampler to +-1 grey level (tests/test_util_cv.py).
"""
import math

import numpy as np

_COEF_BITS = 11
_COEF_SCALE = 1 << _COEF_BITS


def _axis(dn, sn, inv_scale, zero_outside):
    """Source index, its neighbour and the two fixed-point weights for every destination index."""
    d = np.arange(dn, dtype=np.float64)
    f = ((d + 0.5) * (1.0 / inv_scale) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if zero_outside:                       # x: the fraction goes with the clamped index
        lo, hi = s < 0, s >= sn - 1
        f[lo | hi] = 0.0
        s[lo] = 0
        s[hi] = sn - 1
    w1 = np.clip(np.rint(f * np.float32(_COEF_SCALE)), -32768, 32767).astype(np.int32)
    w0 = np.clip(np.rint((np.float32(1.0) - f) * np.float32(_COEF_SCALE)), -32768, 32767).astype(np.int32)
    s0 = np.clip(s, 0, sn - 1)
    s1 = np.clip(s + 1, 0, sn - 1)
    return s0, s1, w0, w1


def resize_linear(img, fx=None, fy=None, dsize=None):
    """``cv2.resize(img, (0, 0), fx=fx, fy=fy)`` — or ``cv2.resize(img, dsize)`` with ``dsize`` =
    (width, height), where the sampling scale is the ratio of the sizes — for a uint8 image [H,W] or
    [H,W,C]."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3):
        raise ValueError("resize_linear takes a uint8 image [H,W] or [H,W,C]")
    flat = img.ndim == 2
    src = img[:, :, None] if flat else img
    sh, sw = src.shape[:2]
    if dsize is not None:
        dw, dh = int(dsize[0]), int(dsize[1])
        fx, fy = dw / float(sw), dh / float(sh)
    else:
        dw, dh = int(np.rint(sw * float(fx))), int(np.rint(sh * float(fy)))
    if dw < 1 or dh < 1:
        raise ValueError("resize_linear: empty destination (%d x %d)" % (dw, dh))
    x0, x1, a0, a1 = _axis(dw, sw, float(fx), True)
    y0, y1, b0, b1 = _axis(dh, sh, float(fy), False)

    def hpass(rows):                       # uint8 [dh, W, C] -> int32 [dh, dw, C], values scaled by 2048
        # (only the pixels that are sampled are widened: a 1280 x 960 frame coming down to 240 x 180
        # touches 2 x 2 of every 5.3 x 5.3 pixels)
        return rows[:, x0].astype(np.int32) * a0[None, :, None] + rows[:, x1].astype(np.int32) * a1[None, :, None]
    s0, s1 = hpass(src[y0]), hpass(src[y1])
    out = (((b0[:, None, None] * (s0 >> 4)) >> 16) + ((b1[:, None, None] * (s1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if flat else out


def resize_img(img, max_size):
    """util/cv.py:7-9: the longer side becomes ``max_size`` (both axes by the same factor)."""
    scale = max_size / float(max(img.shape[0], img.shape[1]))
    return resize_linear(img, scale, scale)


def standard_size(img, h=180, w=240):
    """util/cv.py:12-20: scale so that the image covers h x w, then cut the centre."""
    scale = max(h / img.shape[0], w / img.shape[1])
    img = resize_linear(img, scale, scale)
    top = math.floor((img.shape[0] - h) / 2.0)
    left = math.floor((img.shape[1] - w) / 2.0)
    return img[top:top + h, left:left + w, :]


def merge_images(left_image, right_image):
    """util/cv.py:30-34: the right image scaled to the left one's height, side by side (the example
    pictures of the localisation check, train/train.py:400-420)."""
    right = resize_linear(right_image, dsize=(right_image.shape[1] * left_image.shape[0] // right_image.shape[0],
                                              left_image.shape[0]))
    return np.concatenate((left_image, right), axis=1)


def put_text(text, image, scale=1, color=(0, 255, 0)):
    """util/cv.py:23-27: a caption at the lower-left anchor (10, 35).  The reference draws OpenCV's
    Hershey glyphs; this draws PIL's default font — the pictures are for people, no number depends
    on them."""
    from PIL import Image, ImageDraw, ImageFont
    im = Image.fromarray(np.asarray(image, dtype=np.uint8))
    try:
        font = ImageFont.load_default(size=int(round(28 * scale)))
    except TypeError:                                        # older Pillow: fixed-size bitmap font
        font = ImageFont.load_default()
    draw = ImageDraw.Draw(im)
    draw.text((10, 35), str(text), fill=tuple(int(c) for c in color), font=font, anchor='ls')
    return np.asarray(im)
