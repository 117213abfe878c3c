#!/usr/bin/env python3
"""Deterministic synthetic moving-pattern clip (8-bit 4:2:0 planar YUV).

The clip has fractional-pel global motion, a second object moving at a
different fractional velocity, a brightness ramp (so that bi-prediction and
residuals are exercised) and smooth chroma waves.  Whole-pixel motion is
avoided on purpose: it would yield almost only integer motion vectors and the
8-tap / 4-tap interpolation filters would hardly run.

usage: gen_clip.py WIDTH HEIGHT FRAMES SEED OUT.yuv [noise]
With "noise" a 104x88 patch of white noise (all three planes) is pasted at luma (36, 44) of every frame:
incompressible content, for which an encoder with PCM enabled picks IPCM coding units at low QP.
With "fade" the clip fades towards black (luma scaled by 1 - 0.08 t, chroma amplitude likewise), which makes an
encoder with weighted prediction enabled pick non-trivial weights and offsets.
"""
import sys
import numpy as np


def _box3(t):
    for _ in range(3):
        t = (t + np.roll(t, 1, 0) + np.roll(t, -1, 0)) / 3.0
        t = (t + np.roll(t, 1, 1) + np.roll(t, -1, 1)) / 3.0
    return t


def _bilinear(tex, yy, xx):
    h, w = tex.shape
    y0 = np.floor(yy).astype(np.int64)
    x0 = np.floor(xx).astype(np.int64)
    fy = (yy - y0).astype(np.float32)
    fx = (xx - x0).astype(np.float32)
    y0 %= h
    x0 %= w
    y1 = (y0 + 1) % h
    x1 = (x0 + 1) % w
    return ((1 - fy) * ((1 - fx) * tex[y0, x0] + fx * tex[y0, x1])
            + fy * ((1 - fx) * tex[y1, x0] + fx * tex[y1, x1]))


def frames(width, height, nframes, seed, noise=False, fade=False):
    rng = np.random.default_rng(seed)
    nrng = np.random.default_rng(seed + 1)
    tex = _box3(rng.integers(0, 256, (2 * height, 2 * width)).astype(np.float32))
    # stretch contrast back after the box filters
    tex = (tex - tex.mean()) * 4.0 + 128.0
    y, x = np.mgrid[0:height, 0:width].astype(np.float32)
    scale = 4.5 if width >= 1280 else 1.0
    oh, ow = int(80 * scale), int(120 * scale)
    oy, ox = np.mgrid[0:oh, 0:ow].astype(np.float32)
    cy, cx = np.mgrid[0:height // 2, 0:width // 2].astype(np.float32)
    for t in range(nframes):
        luma = (0.6 * _bilinear(tex, y + 1.5 * t, x + 2.25 * t)
                + 0.4 * (((x + y + 5 * t) / 3.0) % 256.0)) * (1 + 0.004 * t)
        py, px = int(60 * scale / 4.5 + 2 * t), int(100 * scale / 4.5 + 3 * t)
        py, px = min(py, height - oh), min(px, width - ow)
        luma[py:py + oh, px:px + ow] = _bilinear(tex, oy - 0.75 * t + 300, ox - 1.25 * t + 500)
        cb = 128 + 20 * np.sin((cx + 1.5 * t) / 17.0)
        cr = 128 + 30 * np.cos((cy + 0.75 * t) / 11.0)
        if fade:
            g = 1.0 - 0.08 * t
            luma, cb, cr = luma * g + 6.0 * t, 128 + (cb - 128) * g, 128 + (cr - 128) * g
        planes = [np.clip(np.rint(p), 0, 255).astype(np.uint8) for p in (luma, cb, cr)]
        if noise:
            planes[0][44:132, 36:140] = nrng.integers(0, 256, (88, 104))
            planes[1][22:66, 18:70] = nrng.integers(0, 256, (44, 52))
            planes[2][22:66, 18:70] = nrng.integers(0, 256, (44, 52))
        yield tuple(planes)


def main():
    w, h, n, seed = (int(a) for a in sys.argv[1:5])
    with open(sys.argv[5], "wb") as f:
        for planes in frames(w, h, n, seed, noise="noise" in sys.argv[6:], fade="fade" in sys.argv[6:]):
            for p in planes:
                f.write(p.tobytes())


if __name__ == "__main__":
    main()
