"""Global height field assembled from sub-terrain tiles (drop-in for
legged_gym/utils/terrain.py:38-164).  ``height_field_raw`` is int16 [tot_rows, tot_cols] in units
of ``vertical_scale``; ``env_origins[level, type]`` is the spawn point of each tile."""
import numpy as np

from . import terrain_utils


def gap_terrain(terrain, gap_size, platform_size=1.0):
    gap = int(gap_size / terrain.horizontal_scale)
    plat = int(platform_size / terrain.horizontal_scale)
    cx, cy = terrain.length // 2, terrain.width // 2
    x1, y1 = (terrain.length - plat) // 2, (terrain.width - plat) // 2
    x2, y2 = x1 + gap, y1 + gap
    terrain.height_field_raw[cx - x2:cx + x2, cy - y2:cy + y2] = -1000
    terrain.height_field_raw[cx - x1:cx + x1, cy - y1:cy + y1] = 0


def pit_terrain(terrain, depth, platform_size=1.0):
    d = int(depth / terrain.vertical_scale)
    half = int(platform_size / terrain.horizontal_scale / 2)
    x1, x2 = terrain.length // 2 - half, terrain.length // 2 + half
    y1, y2 = terrain.width // 2 - half, terrain.width // 2 + half
    terrain.height_field_raw[x1:x2, y1:y2] = -d


class Terrain:
    def __init__(self, cfg, num_robots) -> None:
        self.cfg, self.num_robots, self.type = cfg, num_robots, cfg.mesh_type
        if self.type in ("none", "plane"):
            return
        self.env_length, self.env_width = cfg.terrain_length, cfg.terrain_width
        self.proportions = [np.sum(cfg.terrain_proportions[:i + 1]) for i in range(len(cfg.terrain_proportions))]
        cfg.num_sub_terrains = cfg.num_rows * cfg.num_cols
        self.env_origins = np.zeros((cfg.num_rows, cfg.num_cols, 3))
        self.width_per_env_pixels = int(self.env_width / cfg.horizontal_scale)
        self.length_per_env_pixels = int(self.env_length / cfg.horizontal_scale)
        self.border = int(cfg.border_size / cfg.horizontal_scale)
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
        if self.type == "trimesh":
            self.vertices, self.triangles = terrain_utils.convert_heightfield_to_trimesh(
                self.height_field_raw, cfg.horizontal_scale, cfg.vertical_scale, cfg.slope_treshold)

    def randomized_terrain(self):
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            choice = np.random.uniform(0, 1)
            difficulty = np.random.choice([0.5, 0.75, 0.9])
            self.add_terrain_to_map(self.make_terrain(choice, difficulty), i, j)

    def curiculum(self):                      # (sic) name kept from the reference API
        for j in range(self.cfg.num_cols):
            for i in range(self.cfg.num_rows):
                self.add_terrain_to_map(
                    self.make_terrain(j / self.cfg.num_cols + 0.001, i / self.cfg.num_rows), i, j)

    def selected_terrain(self):
        kw = dict(self.cfg.terrain_kwargs)
        gen = getattr(terrain_utils, kw.pop("type").split(".")[-1])
        for k in range(self.cfg.num_sub_terrains):
            i, j = np.unravel_index(k, (self.cfg.num_rows, self.cfg.num_cols))
            t = self._blank()
            gen(t, **kw)
            self.add_terrain_to_map(t, i, j)

    def _blank(self):
        return terrain_utils.SubTerrain("terrain", width=self.width_per_env_pixels,
                                        length=self.width_per_env_pixels,
                                        vertical_scale=self.cfg.vertical_scale,
                                        horizontal_scale=self.cfg.horizontal_scale)

    def make_terrain(self, choice, difficulty):
        """One tile: `choice` in [0, 1) picks the generator through the cumulative `terrain_proportions`
        (smooth slope up/down | rough slope | stairs down/up | discrete obstacles | stepping stones | gap | pit),
        `difficulty` in [0, 1) scales it.  Parameter formulas and call order as in the reference (terrain.py:109-145):
        the generators draw from numpy's global stream, so the order is part of the result."""
        tile = self._blank()
        edges = self.proportions
        bucket = next((k for k, edge in enumerate(edges) if choice < edge), len(edges))
        tu = terrain_utils
        if bucket <= 1:
            grade = 0.4 * difficulty
            if bucket == 0 and choice < edges[0] / 2:
                grade = -grade
            tu.pyramid_sloped_terrain(tile, slope=grade, platform_size=3.0)
            if bucket == 1:
                tu.random_uniform_terrain(tile, min_height=-0.05, max_height=0.05, step=0.005, downsampled_scale=0.2)
        elif bucket <= 3:
            rise = 0.05 + 0.18 * difficulty
            tu.pyramid_stairs_terrain(tile, step_width=0.31, step_height=-rise if bucket == 2 else rise, platform_size=3.0)
        elif bucket == 4:
            tu.discrete_obstacles_terrain(tile, 0.05 + 0.2 * difficulty, 1.0, 2.0, 20, platform_size=3.0)
        elif bucket == 5 and len(edges) > 5:
            tu.stepping_stones_terrain(tile, stone_size=1.5 * (1.05 - difficulty), stone_distance=0.05 if difficulty == 0 else 0.1,
                                       max_height=0.0, platform_size=4.0)
        elif bucket == 6 and len(edges) > 6:
            gap_terrain(tile, gap_size=1.0 * difficulty, platform_size=3.0)
        else:
            pit_terrain(tile, depth=1.0 * difficulty, platform_size=4.0)
        return tile

    def add_terrain_to_map(self, terrain, row, col):
        sx = self.border + row * self.length_per_env_pixels
        sy = self.border + col * self.width_per_env_pixels
        self.height_field_raw[sx:sx + self.length_per_env_pixels, sy:sy + self.width_per_env_pixels] = \
            terrain.height_field_raw
        hs = terrain.horizontal_scale
        x1, x2 = int((self.env_length / 2.0 - 1) / hs), int((self.env_length / 2.0 + 1) / hs)
        y1, y2 = int((self.env_width / 2.0 - 1) / hs), int((self.env_width / 2.0 + 1) / hs)
        z = np.max(terrain.height_field_raw[x1:x2, y1:y2]) * terrain.vertical_scale
        self.env_origins[row, col] = [(row + 0.5) * self.env_length, (col + 0.5) * self.env_width, z]
