"""Deterministic input / weight generators shared by make_golden.py (run once, where
/root/reference exists) and the tests (run anywhere).  Keeping inputs and weights as
functions of (name, shape) keeps the committed fixtures small: they hold reference
OUTPUTS (plus small integer inputs), not megabytes of random floats.

Only numpy's legacy MT19937 `RandomState` is used: its stream is frozen by NumPy's
compatibility policy, so the same arrays come out on every box."""
import zlib

import numpy as np


def _seed(name):
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def randn(name, shape, scale=1.0):
    return (np.random.RandomState(_seed(name)).standard_normal(size=tuple(shape)) * scale).astype(np.float32)


def rand(name, shape, lo=0.0, hi=1.0):
    return (np.random.RandomState(_seed(name)).uniform(lo, hi, size=tuple(shape))).astype(np.float32)


def det_state_dict(shapes, scale=None):
    """shapes: {param_name: shape}.  Weights ~ N(0, 0.5/sqrt(fan_in)), 1-D params ~ N(1, .1)
    for norm weights ('weight' of a 1-D shape) and N(0,.1) otherwise; running_var in [0.5,1.5]."""
    out = {}
    for k in sorted(shapes):
        shp = tuple(shapes[k])
        if k.endswith("num_batches_tracked"):
            out[k] = np.zeros(shp, np.int64)
        elif k.endswith("running_var"):
            out[k] = rand(k, shp, 0.5, 1.5)
        elif len(shp) <= 1:
            if k.endswith("weight"):
                out[k] = (1.0 + randn(k, shp, 0.1)).astype(np.float32)
            else:
                out[k] = randn(k, shp, 0.1)
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            out[k] = randn(k, shp, 0.5 / np.sqrt(max(fan_in, 1)))
    return out


def random_voxels(name, batch, shape, n, dup_free=True):
    """n distinct voxel coordinates (b,z,y,x) int32 in a [batch,*shape] grid, in a
    shuffled (non-sorted) order like a voxeliser would emit."""
    rs = np.random.RandomState(_seed(name))
    vol = int(np.prod(shape))
    flat = rs.choice(batch * vol, size=n, replace=False)
    b, r = np.divmod(flat, vol)
    z, r = np.divmod(r, shape[1] * shape[2])
    y, x = np.divmod(r, shape[2])
    return np.stack([b, z, y, x], 1).astype(np.int32)


def clustered_voxels(name, batch, shape, n_seeds, walk):
    """Spatially clustered distinct voxels (random walks from seeds) - closer to LiDAR
    occupancy than uniform noise, so neighbourhoods are well populated."""
    rs = np.random.RandomState(_seed(name))
    pts = set()
    out = []
    for b in range(batch):
        for _ in range(n_seeds):
            p = np.array([rs.randint(0, s) for s in shape])
            for _ in range(walk):
                p = np.clip(p + rs.randint(-1, 2, size=3), 0, np.array(shape) - 1)
                t = (b, int(p[0]), int(p[1]), int(p[2]))
                if t not in pts:
                    pts.add(t)
                    out.append(t)
    arr = np.array(out, np.int32)
    rs.shuffle(arr)
    return arr


def bev_boxes(name, n, spread=10.0, special=True):
    """[n, 7] (x, y, z, dx, dy, dz, heading) boxes clustered enough to overlap; the first rows are special cases
    (identical, axis-aligned touching / nested, 90-degree and 45-degree rotations, a tiny and a huge box)."""
    rs = np.random.RandomState(_seed(name))
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rs.uniform(-spread, spread, (n, 2))
    b[:, 2] = rs.uniform(-2, 1, n)
    b[:, 3] = rs.uniform(0.5, 5.0, n)
    b[:, 4] = rs.uniform(0.5, 2.5, n)
    b[:, 5] = rs.uniform(1.0, 2.0, n)
    b[:, 6] = rs.uniform(-np.pi, np.pi, n)
    if special and n >= 8:
        b[0] = [0, 0, 0, 2, 2, 1, 0]
        b[1] = [0, 0, 0, 2, 2, 1, 0]
        b[2] = [2, 0, 0, 2, 2, 1, 0]
        b[3] = [0.5, 0.5, 0, 2, 2, 1, np.pi / 2]
        b[4] = [0, 0, 0, 2, 2, 1, np.pi / 4]
        b[5] = [0.2, -0.1, 0, 0.05, 0.05, 1, 1.0]
        b[6] = [0, 0, 0, 30, 30, 1, 0.3]
        b[7] = [0, 0, 0, 1, 1, 1, 0]
    return b
