"""Sub-terrain height generators.

The reference builds its terrain out of ``isaacgym.terrain_utils`` (call sites
legged_gym/utils/terrain.py:70,100,110,125-139), a third-party module that is not in the
reference tree and not installable here.  These are fresh restatements
of the generators' published behaviour -- **parity with Isaac Gym is unpinned** (no copy of it
to check against); what *is* pinned is everything the reference's own ``Terrain`` class does on
top of them (tile placement, origins), see tests/test_terrain.py.

All generators work on a ``SubTerrain`` whose ``height_field_raw`` is int16 in units of
``vertical_scale`` metres, indexed [x (width), y (length)] with ``horizontal_scale`` metres per
sample, and modify it in place.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0,
                 horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def _bilinear_resample(coarse: np.ndarray, nx: int, ny: int) -> np.ndarray:
    """Linear interpolation of a coarse grid onto nx x ny samples spanning the same extent."""
    cx, cy = coarse.shape
    fx = np.linspace(0.0, cx - 1.0, nx)
    fy = np.linspace(0.0, cy - 1.0, ny)
    x0 = np.clip(np.floor(fx).astype(int), 0, cx - 2) if cx > 1 else np.zeros(nx, int)
    y0 = np.clip(np.floor(fy).astype(int), 0, cy - 2) if cy > 1 else np.zeros(ny, int)
    tx = (fx - x0)[:, None]
    ty = (fy - y0)[None, :]
    x1 = np.minimum(x0 + 1, cx - 1)
    y1 = np.minimum(y0 + 1, cy - 1)
    c = coarse.astype(np.float64)
    return ((1 - tx) * (1 - ty) * c[np.ix_(x0, y0)] + tx * (1 - ty) * c[np.ix_(x1, y0)]
            + (1 - tx) * ty * c[np.ix_(x0, y1)] + tx * ty * c[np.ix_(x1, y1)])


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Add uniform noise drawn on a coarse grid (``downsampled_scale`` m) and interpolated."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo = int(min_height / terrain.vertical_scale)
    hi = int(max_height / terrain.vertical_scale)
    st = int(step / terrain.vertical_scale)
    levels = np.arange(lo, hi + st, st)
    nx = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    ny = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(levels, (nx, ny))
    fine = np.rint(_bilinear_resample(coarse, terrain.width, terrain.length))
    terrain.height_field_raw += fine.astype(np.int16)
    return terrain


def sloped_terrain(terrain, slope=1):
    x = np.arange(terrain.width).reshape(terrain.width, 1)
    top = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * terrain.width)
    terrain.height_field_raw[:, :] += (top * x / terrain.width).astype(terrain.height_field_raw.dtype)
    return terrain


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.0):
    """Pyramid with the given slope, its tip flattened to a square platform."""
    cx, cy = int(terrain.width / 2), int(terrain.length / 2)
    rx = ((cx - np.abs(cx - np.arange(terrain.width))) / cx).reshape(terrain.width, 1)
    ry = ((cy - np.abs(cy - np.arange(terrain.length))) / cy).reshape(1, terrain.length)
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (terrain.width / 2))
    terrain.height_field_raw += (peak * rx * ry).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = terrain.width // 2 - half, terrain.length // 2 - half
    edge = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(edge, 0), max(edge, 0))
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.0):
    """Concentric square steps rising (or descending for negative height) to a centre platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    x0, x1, y0, y1, h = 0, terrain.width, 0, terrain.length, 0
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0, x1, y0, y1, h = x0 + sw, x1 - sw, y0 + sw, y1 - sw, h + sh
        terrain.height_field_raw[x0:x1, y0:y1] = h
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.0):
    """Random axis-aligned boxes of height {-h, -h/2, h/2, h}; flat platform in the centre."""
    mh = int(max_height / terrain.vertical_scale)
    smin = int(min_size / terrain.horizontal_scale)
    smax = int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    ni, nj = terrain.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(smin, smax, 4)
    for _ in range(num_rects):
        w = np.random.choice(sizes)
        l = np.random.choice(sizes)
        si = np.random.choice(range(0, ni - w, 4))
        sj = np.random.choice(range(0, nj - l, 4))
        terrain.height_field_raw[si:si + w, sj:sj + l] = np.random.choice(heights)
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1.0,
                            depth=-10):
    """Square stones separated by gaps of ``depth`` metres; flat platform in the centre."""
    ss = int(stone_size / terrain.horizontal_scale)
    sd = int(stone_distance / terrain.horizontal_scale)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    heights = np.arange(-mh - 1, mh, step=1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    sx = 0
    while sx < terrain.width:
        ex = min(terrain.width, sx + ss)
        sy = np.random.randint(0, max(ss, 1))
        terrain.height_field_raw[sx:ex, 0:max(0, sy - sd)] = np.random.choice(heights)
        while sy < terrain.length:
            ey = min(terrain.length, sy + ss)
            terrain.height_field_raw[sx:ex, sy:ey] = np.random.choice(heights)
            sy += ss + sd
        sx += ss + sd
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale,
                                   slope_threshold=None):
    """Two triangles per grid cell.  The HIP physics collides against the height samples
    directly, so this exists for API compatibility (``Terrain.vertices/triangles``); the
    ``slope_threshold`` vertical-wall correction of Isaac Gym is not applied."""
    hf = np.asarray(height_field_raw)
    nr, nc = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (nc - 1) * horizontal_scale, nc),
                         np.linspace(0, (nr - 1) * horizontal_scale, nr))
    verts = np.zeros((nr * nc, 3), dtype=np.float32)
    verts[:, 0] = xx.flatten()
    verts[:, 1] = yy.flatten()
    verts[:, 2] = hf.flatten() * vertical_scale
    idx = np.arange(nr * nc).reshape(nr, nc)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    tris = np.concatenate([np.stack([a, d, b], 1), np.stack([a, c, d], 1)]).astype(np.uint32)
    return verts, tris
