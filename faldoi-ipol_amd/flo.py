"""Middlebury .flo codec -- the file format between the reference's three executables.

Layout (reference writer src/iio.c:2539-2553, reader src/iio.c:1807): the 4 bytes "PIEH", int32
width, int32 height (little endian), then height*width interleaved (u, v) float32 pairs.
In memory this package keeps flows planar, shape (2, h, w), like the reference's `u1 | u2` block
(src/energy_model.cpp:167-168).
"""
import struct

import numpy as np

MAGIC = b"PIEH"


def flo_bytes(flow):
    flow = np.asarray(flow, dtype=np.float32)
    if flow.ndim != 3 or flow.shape[0] != 2:
        raise ValueError("flow must have shape (2, h, w)")
    _, h, w = flow.shape
    return MAGIC + struct.pack("<ii", w, h) + np.ascontiguousarray(flow.transpose(1, 2, 0)).astype("<f4").tobytes()


def write_flo(path, flow):
    with open(path, "wb") as f:
        f.write(flo_bytes(flow))


def read_flo(path):
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != MAGIC:
        raise ValueError("%s: not a .flo file (bad magic)" % path)
    w, h = struct.unpack("<ii", data[4:12])
    if w <= 0 or h <= 0 or len(data) != 12 + 8 * w * h:
        raise ValueError("%s: .flo header %dx%d does not match file size" % (path, w, h))
    a = np.frombuffer(data, dtype="<f4", offset=12).reshape(h, w, 2)
    return np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32)


def epe(a, b):
    """Per-pixel end-point error between two planar flows."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)
