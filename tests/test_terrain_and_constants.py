"""Host-side setup pinned against the reference: this repo's Terrain class reproduces the height field
and tile origins the reference's Terrain class built (same generators, same numpy seed), and
draw_env_constants reproduces friction / base-mass / origins / terrain levels draw for draw."""
import numpy as np
import pytest
import torch

from tests import harness

SEEDS = {"anymal_c_flat": 11, "anymal_c_rough": 12, "cassie": 13}


@pytest.mark.parametrize("name", ["anymal_c_rough", "cassie"])
def test_terrain_matches_reference(name):
    from legged_gym_dev_amd.utils.terrain import Terrain
    z, meta = harness.load_fixture(name)
    cfg = harness.make_cfg(name)
    np.random.seed(SEEDS[name])
    t = Terrain(cfg.terrain, cfg.env.num_envs)
    np.testing.assert_array_equal(t.heightsamples, z["const_height_samples"])
    np.testing.assert_allclose(t.env_origins, z["const_terrain_origins"], rtol=0, atol=1e-6)
    assert (t.tot_rows, t.tot_cols) == z["const_height_samples"].shape
    assert t.vertices.shape[0] == t.tot_rows * t.tot_cols and t.triangles.shape[1] == 3


SEEDS.update(anymal_c_randomised=20, anymal_c_flat_curriculum=19)


@pytest.mark.parametrize("name", ["anymal_c_flat", "anymal_c_rough", "cassie", "anymal_c_randomised"])
def test_env_constants_match_reference_draws(name):
    from legged_gym_dev_amd.envs.base.legged_robot import draw_env_constants
    from legged_gym_dev_amd.model.robot_model import resolve_model
    from legged_gym_dev_amd.utils.terrain import Terrain
    z, meta = harness.load_fixture(name)
    cfg = harness.make_cfg(name)
    torch.manual_seed(SEEDS[name])
    np.random.seed(SEEDS[name])
    terrain = Terrain(cfg.terrain, cfg.env.num_envs) if cfg.terrain.mesh_type in ("heightfield", "trimesh") else None
    model = resolve_model("", meta["robot"])
    c = draw_env_constants(cfg, cfg.env.num_envs, float(model["bodies"][0]["mass"]), terrain, num_shapes=model["num_shapes"])
    np.testing.assert_allclose((c["start_pos"] - c["env_origins"]).numpy()[:, :2],
                               z["const_start_xy"] - z["const_env_origins_init"][:, :2], rtol=0, atol=1e-6)   # create_actor's start pose
    if name == "anymal_c_randomised":       # restitution / compliance / thickness per shape, then base mass, then inverse base mass
        assert z["const_shape_props"].shape == (cfg.env.num_envs, model["num_shapes"], 3) and np.abs(z["const_shape_props"]).min() > 0
        np.testing.assert_allclose(c["shape_props"], z["const_shape_props"], rtol=1e-15)
        np.testing.assert_allclose(c["base_inv_mass"], z["const_base_inv_mass"], rtol=1e-15)
        np.testing.assert_allclose(c["material"][:, :3].numpy(), z["const_shape_props"].mean(1), rtol=1e-6)
    np.testing.assert_allclose(c["env_origins"].numpy(), z["const_env_origins_init"], rtol=0, atol=1e-6)
    np.testing.assert_array_equal(c["friction"].numpy(), z["const_friction_coeffs"])
    if cfg.domain_rand.randomize_base_mass:
        np.testing.assert_allclose(c["base_mass"], z["const_base_mass"], rtol=1e-12)
    if terrain is not None:
        np.testing.assert_array_equal(c["terrain_levels"].numpy(), z["const_terrain_levels_init"])
        np.testing.assert_array_equal(c["terrain_types"].numpy(), z["const_terrain_types"])


def test_sharded_constants_are_slices_of_the_global_draw():
    """rank r of G owns envs [r N/G, (r+1) N/G): per-env constants and terrain types use GLOBAL ids."""
    from legged_gym_dev_amd.envs.base.legged_robot import draw_env_constants
    from legged_gym_dev_amd.utils.terrain import Terrain
    cfg = harness.make_cfg("anymal_c_rough")
    np.random.seed(3)
    terrain = Terrain(cfg.terrain, 128)
    outs = []
    for _ in range(2):
        torch.manual_seed(3)
        np.random.seed(4)
        outs.append(draw_env_constants(cfg, 128, 26.0, terrain))
    for k in ("env_origins", "friction", "terrain_levels", "terrain_types"):
        assert torch.equal(outs[0][k], outs[1][k])
    types = outs[0]["terrain_types"]
    assert types.min() == 0 and types.max() == cfg.terrain.num_cols - 1 and (types[1:] >= types[:-1]).all()


def test_randomized_and_selected_terrain_modes():
    """The two non-curriculum modes of Terrain (reference terrain.py:63-66,75-107; not used by the registered tasks, so
    there is no fixture for them): every tile is filled, tile origins sit on the tile centres at the local surface height,
    `selected` applies the named generator to every tile, and a fixed numpy seed reproduces the field."""
    from legged_gym_dev_amd.utils.terrain import Terrain
    cfg = harness.make_cfg("anymal_c_rough").terrain
    cfg.curriculum = False
    np.random.seed(7)
    a = Terrain(cfg, 64)
    np.random.seed(7)
    b = Terrain(cfg, 64)
    assert np.array_equal(a.height_field_raw, b.height_field_raw) and a.height_field_raw.dtype == np.int16
    assert a.height_field_raw.shape == (a.tot_rows, a.tot_cols)
    bd, L, W = a.border, a.length_per_env_pixels, a.width_per_env_pixels
    tiles = [a.height_field_raw[bd + i * L: bd + (i + 1) * L, bd + j * W: bd + (j + 1) * W] for i in range(cfg.num_rows) for j in range(cfg.num_cols)]
    assert sum(int(np.any(t != 0)) for t in tiles) >= len(tiles) - 1            # (a zero-slope pyramid may be flat)
    assert np.all(a.height_field_raw[:bd] == 0) and np.all(a.height_field_raw[:, :bd] == 0)
    for i in range(cfg.num_rows):
        for j in range(cfg.num_cols):
            assert abs(a.env_origins[i, j, 0] - (i + 0.5) * cfg.terrain_length) < 1e-9
            assert abs(a.env_origins[i, j, 1] - (j + 0.5) * cfg.terrain_width) < 1e-9
    cfg.selected = True
    cfg.terrain_kwargs = {"type": "terrain_utils.pyramid_stairs_terrain", "step_width": 0.31, "step_height": 0.1, "platform_size": 3.0}
    s = Terrain(cfg, 64)
    t0 = s.height_field_raw[bd:bd + L, bd:bd + W]
    t1 = s.height_field_raw[bd + L:bd + 2 * L, bd + W:bd + 2 * W]
    assert np.array_equal(t0, t1) and t0.max() > 0                              # the same staircase on every tile
    assert np.allclose(s.env_origins[..., 2], s.env_origins[0, 0, 2]) and s.env_origins[0, 0, 2] > 0.5
