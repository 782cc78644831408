"""Seeded synthetic frames for tests and bench (SURVEY.md §8(d) "Synthetic inputs").

Texture T(seed, W, H): see texture().  Gives >> 10*nfeatures FAST-20 corners over the pyramid
plus FAST-7-only cells and empty cells.  A stream is the same canvas cropped at a moving offset plus +-2 gray uniform noise, the
TUM-mono-like motion BASELINE.json configs 3-5 ask for.

The generator is numpy + a counter-based splitmix64, so every element is a pure function of
(seed, index): identical on every machine, no state to carry to the GPU box.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n, offset=0):
    """n outputs of splitmix64 started at `seed`, elements offset..offset+n-1 (uint64 array)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _box3(img):
    H, W = img.shape
    p = np.pad(img.astype(np.uint16), 1, mode="edge")
    acc = np.zeros((H, W), dtype=np.uint16)
    for dy in range(3):
        for dx in range(3):
            acc += p[dy:dy + H, dx:dx + W]
    return ((acc + 4) // 9).astype(np.uint8)


def texture(seed, W, H, block_sizes=(7, 13, 29, 61), rect_density=0.004):
    """uint8 HxW canvas T(seed, W, H).

    Mean of four uniform-random block layers (block edge 7/13/29/61 px: corners at every pyramid
    scale), overlaid with round(rect_density*W*H) filled rectangles of edge 4..32 px, one 3x3 box
    blur, then a contrast map: a vertical band (x in [0.62W, 0.78W)) keeps 12 % of its contrast
    (FAST-7-only cells, exercises the iniThFAST -> minThFAST fallback) and a 96x96 patch is flat
    (cells with no keypoint at all).  ~3.5 k FAST candidates at level 0 of 640x480, ~16 k per frame.
    """
    acc = np.zeros((H, W), dtype=np.uint32)
    for i, s in enumerate(block_sizes):
        bw, bh = (W + s - 1) // s, (H + s - 1) // s
        r = splitmix64(seed * 1000 + i, bw * bh)
        b = (r % np.uint64(256)).astype(np.uint32).reshape(bh, bw)
        acc += np.kron(b, np.ones((s, s), dtype=np.uint32))[:H, :W]
    img = (acc // len(block_sizes)).astype(np.uint8)
    nrect = int(round(rect_density * W * H))
    rr = splitmix64(seed ^ 0xA5A5A5A5, nrect * 5).reshape(nrect, 5)
    for k in range(nrect):
        rw = 4 + int(rr[k, 0] % np.uint64(29))
        rh = 4 + int(rr[k, 1] % np.uint64(29))
        x0 = int(rr[k, 2] % np.uint64(max(W - 1, 1)))
        y0 = int(rr[k, 3] % np.uint64(max(H - 1, 1)))
        g = int(rr[k, 4] % np.uint64(256))
        img[y0:y0 + rh, x0:x0 + rw] = g
    img = _box3(img)
    # contrast map (integer arithmetic: identical on every platform)
    x0, x1 = (62 * W) // 100, (78 * W) // 100
    band = img[:, x0:x1].astype(np.int32)
    img[:, x0:x1] = (128 + ((band - 128) * 12) // 100).astype(np.uint8)
    fy, fx = (H // 5), (W // 7)
    img[fy:fy + 96, fx:fx + 96] = 90
    return img


def texture_survey(seed, W, H):
    """SURVEY.md 8(d)'s texture as written: 16 x 16-px blocks of uniform-random gray in [16, 240], overlaid with 0.002*W*H filled
    axis-aligned rectangles of edge 8..64 px and random gray, then ONE 3 x 3 box blur.  A realistic corner density (the default
    texture() above is several times denser on purpose): bench.py reports it as the second stream beside the headline one."""
    bw, bh = (W + 15) // 16, (H + 15) // 16
    r = splitmix64(seed * 1000 + 77, bw * bh)
    b = (np.uint64(16) + r % np.uint64(225)).astype(np.uint8).reshape(bh, bw)
    img = np.kron(b, np.ones((16, 16), dtype=np.uint8))[:H, :W].copy()
    nrect = int(round(0.002 * W * H))
    rr = splitmix64(seed ^ 0x5A5A5A5A, nrect * 5).reshape(nrect, 5)
    for k in range(nrect):
        rw = 8 + int(rr[k, 0] % np.uint64(57))
        rh = 8 + int(rr[k, 1] % np.uint64(57))
        x0 = int(rr[k, 2] % np.uint64(max(W - 1, 1)))
        y0 = int(rr[k, 3] % np.uint64(max(H - 1, 1)))
        img[y0:y0 + rh, x0:x0 + rw] = int(np.uint64(16) + rr[k, 4] % np.uint64(225))
    return _box3(img)


def stream_survey(seed, W, H, nframes, step=(2, 1), noise_seed0=100, noise=2, first=0, count=None):
    """stream() on texture_survey(): SURVEY.md 8(d) C4 as written (the canvas shifted by (2k, k) px, noise seed 100 + k)."""
    return stream(seed, W, H, nframes, step, noise_seed0, noise, first, count, canvas_fn=texture_survey)


def stream(seed, W, H, nframes, step=(2, 1), noise_seed0=100, noise=2, first=0, count=None, canvas_fn=None):
    """nframes x H x W uint8: canvas T(seed) cropped at offset k*step plus +-noise gray.
    first/count: only frames [first, first+count) of that nframes-long stream (a rank's shard of it)."""
    pad_x, pad_y = step[0] * (nframes - 1), step[1] * (nframes - 1)
    canvas = (canvas_fn or texture)(seed, W + pad_x, H + pad_y)
    count = nframes - first if count is None else count
    out = np.empty((count, H, W), dtype=np.uint8)
    for j in range(count):
        k = first + j
        ox, oy = step[0] * k, step[1] * k
        crop = canvas[oy:oy + H, ox:ox + W].astype(np.int16)
        if noise > 0:
            r = splitmix64(noise_seed0 + k, W * H)
            crop = crop + (r % np.uint64(2 * noise + 1)).astype(np.int16).reshape(H, W) - noise
        out[j] = np.clip(crop, 0, 255).astype(np.uint8)
    return out


def frame_pair(seed, W, H, shift=(7, 3), noise_seed=3, noise=2):
    """Two frames: T(seed) and the same texture shifted by `shift` px with +-noise gray (config 3)."""
    canvas = texture(seed, W + shift[0], H + shift[1])
    f0 = canvas[:H, :W].copy()
    f1 = canvas[shift[1]:shift[1] + H, shift[0]:shift[0] + W].astype(np.int16)
    r = splitmix64(noise_seed, W * H)
    f1 = f1 + (r % np.uint64(2 * noise + 1)).astype(np.int16).reshape(H, W) - noise
    return f0, np.clip(f1, 0, 255).astype(np.uint8)


def stream_layers(seed, W, H, nframes, shifts=(1, 2, 3), cell=96, noise_seed0=100, noise=2):
    """A non-planar scene for the pose solvers: the image plane is tiled with `cell`-px squares, each
    looking at one of len(shifts) fronto-parallel textured layers; layer r slides shifts[r] px per frame to the left,
    which is what a camera translating along +x by b per frame sees of a plane at depth fx*b/shifts[r].
    Returns (frames nframes x H x W uint8, layer H x W uint8 = index into shifts).  Pixels next to a tile edge see a
    layer change and act as occlusions."""
    nl = len(shifts)
    canv = [texture(seed * 16 + r, W + shifts[r] * (nframes - 1), H) for r in range(nl)]
    gy, gx = np.mgrid[0:H, 0:W]
    pick = splitmix64(seed + 7777, ((H + cell - 1) // cell) * ((W + cell - 1) // cell))
    layer = (pick % np.uint64(nl)).astype(np.uint8).reshape((H + cell - 1) // cell, (W + cell - 1) // cell)[gy // cell, gx // cell]
    out = np.empty((nframes, H, W), dtype=np.uint8)
    for k in range(nframes):
        img = np.zeros((H, W), np.int16)
        for r in range(nl):
            ox = shifts[r] * k
            img = np.where(layer == r, canv[r][:, ox:ox + W].astype(np.int16), img)
        if noise > 0:
            rr = splitmix64(noise_seed0 + k, W * H)
            img = img + (rr % np.uint64(2 * noise + 1)).astype(np.int16).reshape(H, W) - noise
        out[k] = np.clip(img, 0, 255).astype(np.uint8)
    return out, layer
