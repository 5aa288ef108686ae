"""Curriculum terrain grid -> int16 height field (+ per-tile env origins).

`Terrain` mirrors reference legged_gym/utils/terrain.py:38-164 (grid layout, difficulty ramps, origin
rule).  The sub-terrain generators stand in for `isaacgym.terrain_utils` (closed package, absent):
they are restated from that module's documented behaviour — PARITY UNPINNED,
no reference artefact exists for them.  numpy's global RNG is used exactly where the reference does,
so `set_seed` makes the map reproducible.

The height field is what the HIP step kernel collides against.  For mesh_type 'trimesh' the reference converts it
to a triangle mesh whose steep steps become vertical faces (terrain.py:69-73, `slope_treshold`); the kernel
evaluates that same correction per grid cell straight from the height field (csrc/pf_step_kernel.h,
`terrain_query`), so no mesh is needed at run time.  `Terrain.vertices` / `.triangles` are still provided, built on
first access by `convert_heightfield_to_trimesh` below, for callers that export or inspect the mesh.
"""
import numpy as np


class SubTerrain:
    def __init__(self, terrain_name="terrain", width=256, length=256, vertical_scale=1.0, horizontal_scale=1.0):
        self.terrain_name = terrain_name
        self.vertical_scale = vertical_scale
        self.horizontal_scale = horizontal_scale
        self.width = width
        self.length = length
        self.height_field_raw = np.zeros((self.width, self.length), dtype=np.int16)


def pyramid_sloped_terrain(terrain, slope=1, platform_size=1.):
    """Separable tent scaled to int(slope * hs/vs * width/2), clipped at the central platform's height."""
    w, l = terrain.width, terrain.length
    cx, cy = int(w / 2), int(l / 2)
    tx = ((cx - np.abs(cx - np.arange(w))) / cx).reshape(w, 1)
    ty = ((cy - np.abs(cy - np.arange(l))) / cy).reshape(1, l)
    peak = int(slope * (terrain.horizontal_scale / terrain.vertical_scale) * (w / 2))
    terrain.height_field_raw += (peak * tx * ty).astype(terrain.height_field_raw.dtype)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, y1 = w // 2 - half, l // 2 - half
    ref = terrain.height_field_raw[x1, y1]
    terrain.height_field_raw = np.clip(terrain.height_field_raw, min(ref, 0), max(ref, 0))
    return terrain


def random_uniform_terrain(terrain, min_height, max_height, step=1, downsampled_scale=None):
    """Uniform random heights on a coarse grid, bilinearly up-sampled, rounded, added."""
    if downsampled_scale is None:
        downsampled_scale = terrain.horizontal_scale
    lo, hi = int(min_height / terrain.vertical_scale), int(max_height / terrain.vertical_scale)
    st = int(step / terrain.vertical_scale)
    choices = np.arange(lo, hi + st, st)
    nx = int(terrain.width * terrain.horizontal_scale / downsampled_scale)
    ny = int(terrain.length * terrain.horizontal_scale / downsampled_scale)
    coarse = np.random.choice(choices, (nx, ny)).astype(np.float64)
    # bilinear interpolation from the coarse nodes (spanning the whole tile) to the fine grid
    fx = np.linspace(0, nx - 1, terrain.width)
    fy = np.linspace(0, ny - 1, terrain.length)
    ix = np.clip(np.floor(fx).astype(int), 0, nx - 2)
    iy = np.clip(np.floor(fy).astype(int), 0, ny - 2)
    ax = (fx - ix).reshape(-1, 1)
    ay = (fy - iy).reshape(1, -1)
    c00 = coarse[np.ix_(ix, iy)]
    c10 = coarse[np.ix_(ix + 1, iy)]
    c01 = coarse[np.ix_(ix, iy + 1)]
    c11 = coarse[np.ix_(ix + 1, iy + 1)]
    fine = (1 - ax) * ((1 - ay) * c00 + ay * c01) + ax * ((1 - ay) * c10 + ay * c11)
    terrain.height_field_raw += np.rint(fine).astype(np.int16)
    return terrain


def pyramid_stairs_terrain(terrain, step_width, step_height, platform_size=1.):
    """Concentric square steps until the inner square is no larger than the platform."""
    sw = int(step_width / terrain.horizontal_scale)
    sh = int(step_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    x0, x1, y0, y1, h = 0, terrain.width, 0, terrain.length, 0
    while (x1 - x0) > plat and (y1 - y0) > plat:
        x0 += sw; x1 -= sw; y0 += sw; y1 -= sw
        h += sh
        terrain.height_field_raw[x0:x1, y0:y1] = h
    return terrain


def discrete_obstacles_terrain(terrain, max_height, min_size, max_size, num_rects, platform_size=1.):
    """Random rectangles with heights in {-h, -h/2, h/2, h}; flat central platform."""
    mh = int(max_height / terrain.vertical_scale)
    lo, hi = int(min_size / terrain.horizontal_scale), int(max_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    rows, cols = terrain.height_field_raw.shape
    heights = [-mh, -mh // 2, mh // 2, mh]
    sizes = range(lo, hi, 4)
    for _ in range(num_rects):
        wdt = np.random.choice(sizes)
        lng = np.random.choice(sizes)
        si = np.random.choice(range(0, rows - wdt, 4))
        sj = np.random.choice(range(0, cols - lng, 4))
        terrain.height_field_raw[si:si + wdt, sj:sj + lng] = np.random.choice(heights)
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def stepping_stones_terrain(terrain, stone_size, stone_distance, max_height, platform_size=1., depth=-10):
    """Grid of square stones separated by gaps of `depth`; flat central platform."""
    ss = max(int(stone_size / terrain.horizontal_scale), 1)
    sd = max(int(stone_distance / terrain.horizontal_scale), 1)
    mh = int(max_height / terrain.vertical_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    hr = np.arange(-mh - 1, mh, step=1)
    terrain.height_field_raw[:, :] = int(depth / terrain.vertical_scale)
    sx = 0
    while sx < terrain.width:
        sy = 0
        while sy < terrain.length:
            terrain.height_field_raw[sx:sx + ss, sy:sy + ss] = np.random.choice(hr)
            sy += ss + sd
        sx += ss + sd
    x1, x2 = (terrain.width - plat) // 2, (terrain.width + plat) // 2
    y1, y2 = (terrain.length - plat) // 2, (terrain.length + plat) // 2
    terrain.height_field_raw[x1:x2, y1:y2] = 0
    return terrain


def gap_terrain(terrain, gap_size, platform_size=1.):          # reference terrain.py:166-178
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1 = (terrain.length - plat) // 2
    y1 = (terrain.width - plat) // 2
    x2, y2 = x1 + gap, y1 + gap
    terrain.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    terrain.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.):             # reference terrain.py:180-187
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2 = terrain.length // 2 - half, terrain.length // 2 + half
    y1, y2 = terrain.width // 2 - half, terrain.width // 2 + half
    terrain.height_field_raw[x1:x2, y1:y2] = -d


def convert_heightfield_to_trimesh(height_field_raw, horizontal_scale, vertical_scale, slope_threshold=None):
    """Grid -> (vertices float32 (R*C, 3), triangles uint32 (2 (R-1)(C-1), 3)); stands in for the function of that name in
    isaacgym.terrain_utils (call site terrain.py:70; parity unpinned).  Where the height difference between two grid
    neighbours exceeds slope_threshold * horizontal_scale / vertical_scale, the LOW vertex is slid one grid step towards
    the high one (x, y, or -- where neither applies -- diagonally), so the step becomes a vertical face on the high
    vertex's grid line and the ground next to it keeps the low height."""
    hf = np.asarray(height_field_raw)
    R, Cn = hf.shape
    yy, xx = np.meshgrid(np.linspace(0, (Cn - 1) * horizontal_scale, Cn), np.linspace(0, (R - 1) * horizontal_scale, R))
    if slope_threshold is not None:
        thr = slope_threshold * horizontal_scale / vertical_scale
        h = hf.astype(np.int64)
        mx, my, mc = np.zeros((R, Cn)), np.zeros((R, Cn)), np.zeros((R, Cn))
        dx, dy, dd = h[1:, :] - h[:-1, :], h[:, 1:] - h[:, :-1], h[1:, 1:] - h[:-1, :-1]
        mx[:-1, :] += dx > thr            # low vertex before a rise: forward
        mx[1:, :] -= -dx > thr            # low vertex after a drop: back
        my[:, :-1] += dy > thr
        my[:, 1:] -= -dy > thr
        mc[:-1, :-1] += dd > thr
        mc[1:, 1:] -= -dd > thr
        xx = xx + (mx + mc * (mx == 0)) * horizontal_scale
        yy = yy + (my + mc * (my == 0)) * horizontal_scale
    vertices = np.stack([xx.ravel(), yy.ravel(), hf.ravel() * vertical_scale], axis=1).astype(np.float32)
    i0 = (np.arange(R - 1)[:, None] * Cn + np.arange(Cn - 1)[None, :]).ravel()
    tri = np.empty((2 * i0.size, 3), dtype=np.uint32)
    tri[0::2] = np.stack([i0, i0 + Cn + 1, i0 + 1], axis=1)
    tri[1::2] = np.stack([i0, i0 + Cn, i0 + Cn + 1], axis=1)
    return vertices, tri


class Terrain:
    @property
    def vertices(self):
        self._ensure_mesh()
        return self._mesh[0]

    @property
    def triangles(self):
        self._ensure_mesh()
        return self._mesh[1]

    def _ensure_mesh(self):
        """reference terrain.py:68-73 builds the mesh eagerly for mesh_type 'trimesh' (2.7 M vertices); here on first use"""
        if getattr(self, "_mesh", None) is None:
            if self.type != "trimesh":
                raise AttributeError("vertices / triangles exist for mesh_type 'trimesh' only")
            self._mesh = convert_heightfield_to_trimesh(self.height_field_raw, self.cfg.horizontal_scale, self.cfg.vertical_scale,
                                                        self.cfg.slope_treshold)

    def __init__(self, cfg, num_robots) -> None:
        self.cfg = cfg
        self.num_robots = num_robots
        self.type = cfg.mesh_type
        if self.type in ["none", 'plane']:
            return
        self.env_length = cfg.terrain_length
        self.env_width = cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        self.cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / self.cfg.horizontal_scale)
        self.tot_cols = int(cfg.num_cols * self.width_per_env_pixels) + 2 * self.border
        self.tot_rows = int(cfg.num_rows * self.length_per_env_pixels) + 2 * self.border
        self.height_field_raw = np.zeros((self.tot_rows, self.tot_cols), dtype=np.int16)
        if cfg.curriculum:
            self.curiculum()
        elif cfg.selected:
            self.selected_terrain()
        else:
            self.randomized_terrain()
        self.heightsamples = self.height_field_raw

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                difficulty = i / self.cfg.num_rows
                choice = j / self.cfg.num_cols + 0.001
                self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def selected_terrain(self):
        kwargs = dict(self.cfg.terrain_kwargs)
        gen = globals()[kwargs.pop('type').split('.')[-1]]
        for k in range(self.cfg.num_sub_terrains):
            (i, j) = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            terrain = self._blank()
            gen(terrain, **kwargs)
            self.add_terrain_to_map(terrain, i, j)

    def _blank(self):
        return SubTerrain("terrain", width=self.width_per_env_pixels, length=self.width_per_env_pixels,
                          vertical_scale=self.cfg.vertical_scale, horizontal_scale=self.cfg.horizontal_scale)

    def make_terrain(self, choice, difficulty):
        terrain = self._blank()
        slope = difficulty * 0.4
        step_height = 0.05 + 0.18 * difficulty
        discrete_obstacles_height = 0.05 + difficulty * 0.2
        stepping_stones_size = 1.5 * (1.05 - difficulty)
        stone_distance = 0.05 if difficulty == 0 else 0.1
        p = self.proportions
        if choice < p[0]:
            if choice < p[0] / 2:
                slope *= -1
            pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.)
        elif choice < p[1]:
            pyramid_sloped_terrain(terrain, slope=slope, platform_size=3.)
            random_uniform_terrain(terrain, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif choice < p[3]:
            if choice < p[2]:
                step_height *= -1
            pyramid_stairs_terrain(terrain, step_width=0.31, step_height=step_height, platform_size=3.)
        elif choice < p[4]:
            discrete_obstacles_terrain(terrain, discrete_obstacles_height, 1., 2., 20, platform_size=3.)
        elif len(p) > 5 and choice < p[5]:
            stepping_stones_terrain(terrain, stone_size=stepping_stones_size, stone_distance=stone_distance,
                                    max_height=0., platform_size=4.)
        elif len(p) > 6 and choice < p[6]:
            gap_terrain(terrain, gap_size=1. * difficulty, platform_size=3.)
        else:
            pit_terrain(terrain, depth=1. * difficulty, platform_size=4.)
        return terrain

    def add_terrain_to_map(self, terrain, row, col):
        i, j = row, col
        sx, ex = self.border + i * self.length_per_env_pixels, self.border + (i + 1) * self.length_per_env_pixels
        sy, ey = self.border + j * self.width_per_env_pixels, self.border + (j + 1) * self.width_per_env_pixels
        self.height_field_raw[sx:ex, sy:ey] = terrain.height_field_raw
        x1 = int((self.env_length / 2. - 1) / terrain.horizontal_scale)
        x2 = int((self.env_length / 2. + 1) / terrain.horizontal_scale)
        y1 = int((self.env_width / 2. - 1) / terrain.horizontal_scale)
        y2 = int((self.env_width / 2. + 1) / terrain.horizontal_scale)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[i, j] = [(i + 0.5) * self.env_length, (j + 0.5) * self.env_width, z]
