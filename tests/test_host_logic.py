"""CPU tests of the host side: C-ABI export surface, config/registry/CLI mirrors, terrain generator."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
from pointfoot_amd import _abi


def test_library_exports_every_declared_symbol_and_abi_sizes_match():
    hdr = open(os.path.join(common.ROOT, "include", "pointfoot_hip.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b((?:pf|ppo)_[a-z_0-9]+)\s*\(", body))
    assert declared == set(_abi.EXPORTED_SYMBOLS), declared ^ set(_abi.EXPORTED_SYMBOLS)
    lib = C.CDLL(_abi.lib_path())
    for name in declared:
        assert hasattr(lib, name), f"libpointfoot_hip.so does not export {name}"
    _abi.load()      # checks pf_abi_sizes against the ctypes mirror
    # ... and nothing else: every pf_* / ppo_* symbol the library exports is declared in one of the two headers
    import subprocess
    dev = re.sub(r"/\*.*?\*/", "", open(os.path.join(common.ROOT, "include", "pointfoot_hip_dev.h")).read(), flags=re.S)
    dev_declared = set(re.findall(r"\b((?:pf|ppo)_[a-z_0-9]+)\s*\(", dev))
    nm = subprocess.run(["nm", "-D", "--defined-only", _abi.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if re.match(r"^[0-9a-f]+ T (pf|ppo)_", ln)}
    assert exported == declared | dev_declared, exported ^ (declared | dev_declared)


def test_pf_layout_host_only_and_error_reporting():
    lib = _abi.load()
    sim = common.HostSim(common.make_cfg("rough", 8), 8, hf=np.zeros((1300, 2100), np.int16),
                         terrain_origins=np.zeros((10, 20, 3), np.float32))
    L = _abi.PfLayout()
    assert lib.pf_layout(C.byref(sim.model), C.byref(sim.c), C.byref(L)) == 0
    assert bytes(L) == bytes(sim.L) and L.rec_floats % 16 == 0 and L.rand_floats >= 27 + 121 + 18
    Le = _abi.PfLayout()      # the host build of the step program (tests/emu) lays the record out the same way
    assert common.emu_lib().pfe_layout(C.byref(sim.model), C.byref(sim.c), C.byref(Le)) == 0 and bytes(Le) == bytes(L)
    for task in ("flat", "anymal_c_rough", "anymal_c_flat"):
        s2 = common.HostSim(common.make_cfg(task, 4), 4)
        assert common.emu_lib().pfe_layout(C.byref(s2.model), C.byref(s2.c), C.byref(Le)) == 0 and bytes(Le) == bytes(s2.L), task
    assert lib.pf_layout(None, None, None) != 0 and b"null" in lib.pf_last_error()
    bad = _abi.PfModel.from_buffer_copy(bytes(sim.model)); bad.nb = 7
    assert not lib.pf_create(C.byref(bad), C.byref(sim.c), 0)
    assert b"kernel instance" in lib.pf_last_error()


def test_registry_cli_and_config_mirrors(tmp_path):
    from pointfoot_amd.envs import task_registry, PointFoot, PointFootRoughCfg
    from pointfoot_amd.utils.helpers import get_args, class_to_dict, update_cfg_from_args, get_load_path, parse_sim_params
    assert set(task_registry.task_classes) == {"pointfoot_rough", "pointfoot_flat", "anymal_c_rough", "anymal_c_flat", "anymal_b", "a1"}
    with pytest.raises(ValueError, match="was not registered"):
        task_registry.make_env("cassie", args=get_args([]))
    env_cfg, train_cfg = task_registry.get_cfgs("pointfoot_rough")
    assert env_cfg.seed == train_cfg.seed == 1 and isinstance(env_cfg, PointFootRoughCfg)
    d = class_to_dict(train_cfg)
    assert list(d["algorithm"].keys()) == sorted(d["algorithm"].keys()) and "init_member_classes" in d
    assert d["policy"]["actor_hidden_dims"] == [512, 256, 128] and d["runner"]["num_steps_per_env"] == 24
    a = get_args([])
    assert a.task == "pointfoot_rough" and a.sim_device == "cuda:0" and a.rl_device == "cuda:0" and a.physics_engine == 1
    a = get_args(["--task", "pointfoot_flat", "--num_envs", "64", "--seed", "7", "--max_iterations", "3", "--headless", "--resume",
                  "--sim_device", "cpu"])
    assert a.sim_device == "cpu" and not a.use_gpu_pipeline and not a.use_gpu
    import copy
    e2, t2 = update_cfg_from_args(copy.deepcopy(env_cfg), copy.deepcopy(train_cfg), a)
    assert e2.env.num_envs == 64 and t2.seed == 7 and t2.runner.max_iterations == 3 and t2.runner.resume
    sp = parse_sim_params(a, {"sim": class_to_dict(env_cfg.sim)})
    assert sp.dt == 0.005 and sp.physx.num_position_iterations == 4 and sp.use_gpu_pipeline is False
    # get_load_path: last run lexicographically, highest model number (helpers.py:103-125)
    for run, models in (("Jan01_00-00-00_", ["model_0.pt", "model_50.pt", "model_100.pt"]), ("Jan02_00-00-00_", ["model_0.pt", "model_9.pt"])):
        os.makedirs(tmp_path / run)
        for m in models:
            (tmp_path / run / m).write_bytes(b"")
    os.makedirs(tmp_path / "exported")
    assert get_load_path(str(tmp_path)).endswith(os.path.join("Jan02_00-00-00_", "model_9.pt"))
    assert get_load_path(str(tmp_path), load_run="Jan01_00-00-00_").endswith("model_100.pt")
    assert get_load_path(str(tmp_path), checkpoint=50).endswith(os.path.join("Jan02_00-00-00_", "model_50.pt"))
    with pytest.raises(ValueError, match="No runs"):
        get_load_path(str(tmp_path / "nope"))
    # the env refuses to run without the HIP device instead of silently falling back
    with pytest.raises(RuntimeError, match="HIP kernel only|no GPU"):
        PointFoot(PointFootRoughCfg(), sp, 1, "cpu", True)


def test_terrain_generator_layout_and_origins():
    from pointfoot_amd.utils.terrain import Terrain
    from pointfoot_amd.envs import PointFootRoughCfg
    cfg = PointFootRoughCfg()
    np.random.seed(1)
    t = Terrain(cfg.terrain, 4096)
    assert t.height_field_raw.shape == (1300, 2100) and t.height_field_raw.dtype == np.int16
    assert t.env_origins.shape == (10, 20, 3)
    np.testing.assert_allclose(t.env_origins[3, 5, :2], [3.5 * 8, 5.5 * 8])
    hf = t.height_field_raw
    assert (hf[:250] == 0).all() and (hf[:, :250] == 0).all() and (hf[-250:] == 0).all()      # flat 25 m border
    tile = lambda i, j: hf[250 + 80 * i:250 + 80 * (i + 1), 250 + 80 * j:250 + 80 * (j + 1)]
    assert tile(9, 0).min() < 0 and tile(9, 1).max() > 0               # col 0: downward pyramid, col 1: upward
    assert tile(0, 0).min() == 0 and tile(0, 0).max() == 0             # difficulty 0 slope is flat
    st = tile(9, 12)                                                   # stairs up, level 9: step 0.05 + 0.18*0.9
    step = int((0.05 + 0.18 * 0.9) / 0.005)
    assert set(np.unique(st)) <= {k * step for k in range(0, 20)} and st.max() >= 5 * step
    assert tile(9, 5).min() < 0 and tile(9, 5).max() == 0 or tile(9, 5).min() < 0          # stairs down
    assert abs(t.env_origins[9, 12, 2] - st[30:50, 30:50].max() * 0.005) < 1e-9
    np.random.seed(1)
    t2 = Terrain(PointFootRoughCfg().terrain, 4096)
    assert (t2.height_field_raw == hf).all()                            # numpy-seeded, reproducible
    cfg_plane = PointFootRoughCfg()
    cfg_plane.terrain.mesh_type = "plane"
    assert not hasattr(Terrain(cfg_plane.terrain, 8), "height_field_raw")


def test_legged_robot_cfg_defaults_kept():
    from pointfoot_amd.envs import LeggedRobotCfg, LeggedRobotCfgPPO
    c, p = LeggedRobotCfg(), LeggedRobotCfgPPO()
    assert c.env.num_envs == 4096 and c.env.num_observations == 235 and len(c.terrain.measured_points_x) == 17
    assert c.rewards.only_positive_rewards and c.domain_rand.push_interval_s == 15 and c.sim.physx.solver_type == 1
    assert p.algorithm.num_learning_epochs == 5 and p.runner.max_iterations == 1500 and p.policy.activation == "elu"


def test_terrain_grid_logic_matches_reference_fixture(monkeypatch):
    """a15: `pointfoot_amd.utils.terrain.Terrain` against the reference's own `Terrain` (legged_gym/utils/terrain.py:38-164)
    run on the same generators (tests/golden/make_golden.py::terrain_grid_vectors): identical generator call sequence
    (tile order, generator choice, slope / step height / obstacle height per tile = curiculum() + make_terrain()),
    identical pasted field (sha256 over 1300 x 2100 int16) and identical env_origins (add_terrain_to_map's origin rule)."""
    import hashlib
    from pointfoot_amd.utils import terrain as mine
    from pointfoot_amd.envs.pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg
    g = common.load_golden("terrain_grid")
    calls = []
    for name in ("pyramid_sloped_terrain", "random_uniform_terrain", "pyramid_stairs_terrain", "discrete_obstacles_terrain",
                 "stepping_stones_terrain"):
        fn = getattr(mine, name)

        def logged(terrain, *a, _fn=fn, _name=name, **kw):
            calls.append((_name, tuple(float(v) for v in a), tuple(sorted((k, float(v)) for k, v in kw.items()))))
            return _fn(terrain, *a, **kw)
        monkeypatch.setattr(mine, name, logged)
    np.random.seed(int(g["seed"]))
    t = mine.Terrain(PointFootRoughCfg.terrain(), 4096)
    hf = np.ascontiguousarray(t.height_field_raw)
    assert tuple(g["shape"]) == hf.shape == (int(g["tot_rows"]), int(g["tot_cols"])) and t.border == int(g["border_px"])
    names = list(g["call_names"])
    assert [names.index(c[0]) for c in calls] == list(g["call_gen"]), "generator sequence differs from the reference's make_terrain"
    args = np.array([list(c[1]) + [v for _, v in c[2]] + [np.nan] * (8 - len(c[1]) - len(c[2])) for c in calls])
    np.testing.assert_array_equal(args, g["call_args"])
    np.testing.assert_array_equal(hf[::13, ::17], g["coarse"])
    assert hashlib.sha256(hf.tobytes()).digest() == bytes(g["sha256"])
    np.testing.assert_array_equal(t.env_origins, g["env_origins"])


def test_trimesh_conversion_makes_vertical_faces():
    """f1: the mesh restatement (isaacgym.terrain_utils.convert_heightfield_to_trimesh, call site terrain.py:69-73;
    parity unpinned) turns a step steeper than slope_treshold into a vertical face on the high vertex's grid line and
    leaves a gentle slope alone -- the geometry the step kernel's `terrain_query` evaluates per cell."""
    from pointfoot_amd.utils.terrain import convert_heightfield_to_trimesh
    hf = np.zeros((8, 6), np.int16)
    hf[4:, :] = 40            # 0.2 m step between rows 3 and 4 (slope 2.0 > 0.75)
    hf[:, 4:] += 10           # 0.05 m step between columns 3 and 4 (slope 0.5: stays a ramp)
    v, t = convert_heightfield_to_trimesh(hf, 0.1, 0.005, 0.75)
    v = v.reshape(8, 6, 3)
    assert t.shape == (2 * 7 * 5, 3) and t.max() == 8 * 6 - 1
    np.testing.assert_allclose(v[3, :, 0], 0.4, atol=1e-6)        # low row 3 slid under high row 4: face at x = 0.4
    np.testing.assert_allclose(v[4, :, 0], 0.4, atol=1e-6)
    np.testing.assert_allclose(v[2, :, 0], 0.2, atol=1e-6)        # its neighbour stays: the low cell is stretched
    np.testing.assert_allclose(np.delete(v[:, 3, 1], 3), 0.3, atol=1e-6)   # gentle step: untouched (row 3 also slides diagonally: 0.25 m corner rise)
    np.testing.assert_allclose(v[:, :, 2], hf * 0.005, atol=1e-7)
    v0, _ = convert_heightfield_to_trimesh(hf, 0.1, 0.005, None)
    np.testing.assert_allclose(v0.reshape(8, 6, 3)[3, :, 0], 0.3, atol=1e-6)


def test_pmc_traffic_sums_requests_by_size(tmp_path):
    """tools/pmc_traffic.py: with the request-size passes the traffic is 128 B x RDREQ_128B + 64 B x RDREQ_64B + 32 B x RDREQ_32B read and
    64 B x WRREQ_64B + 32 B x the other write requests (profiles/r4_fetch_size_gather_calibration.txt), the doubling-rule figure kept beside it."""
    import csv, json, subprocess, sys
    kern = "void pf_step_kernel<9, 6, 2>(void const*)"

    def pass_dir(name, counters):
        d = tmp_path / name / "x"
        d.mkdir(parents=True)
        with open(d / "1_counter_collection.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for i in range(8):
                for c, v in counters.items():
                    w.writerow([i, kern, c, v])
                w.writerow([i, "other_kernel()", next(iter(counters)), 1e9])
        return str(tmp_path / name)
    dirs = [pass_dir("f", {"FETCH_SIZE": 100.0}), pass_dir("w", {"WRITE_SIZE": 50.0}),
            pass_dir("r1", {"TCC_EA0_RDREQ_128B_sum": 1000.0, "TCC_EA0_RDREQ_64B_sum": 600.0}),
            pass_dir("r2", {"TCC_EA0_RDREQ_sum": 1600.0, "TCC_EA0_RDREQ_32B_sum": 0.0}),
            pass_dir("r3", {"TCC_EA0_WRREQ_sum": 900.0, "TCC_EA0_WRREQ_64B_sum": 700.0})]
    out = tmp_path / "t.json"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), dirs[0], dirs[1], "pf_step_kernel<9", "pointfoot_rough", "4096",
                    str(out)] + dirs[2:], check=True, capture_output=True)
    rec = json.load(open(out))
    assert rec["read_bytes_per_launch"] == 128 * 1000 + 64 * 600 and rec["write_bytes_per_launch"] == 64 * 700 + 32 * 200
    assert rec["traffic_bytes_per_launch"] == rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]
    assert rec["traffic_doubling_rule_bytes_per_launch"] == (2 * 100.0 + 50.0) * 1024
    subprocess.run([sys.executable, os.path.join(root, "tools", "pmc_traffic.py"), dirs[0], dirs[1], "pf_step_kernel<9", "pointfoot_rough", "4096",
                    str(out)], check=True, capture_output=True)        # without the request-size passes: the guide's doubling rule
    assert json.load(open(out))["traffic_bytes_per_launch"] == (2 * 100.0 + 50.0) * 1024
