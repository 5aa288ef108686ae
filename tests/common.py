"""Shared test plumbing: builds/loads the CPU oracle and the host-emulation build of the HIP step
program, and drives either (or the real HIP library) through the same PfBuffers/PfInject structs on the
golden fixtures.  Test infrastructure only."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from pointfoot_amd import _abi  # noqa: E402
from pointfoot_amd.envs.pointfoot import _native  # noqa: E402
from pointfoot_amd.utils.helpers import SimParams  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def _build(cmd, out, srcs):
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in srcs):
        return
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(cmd, cwd=ROOT)


def oracle_lib():
    out = os.path.join(ROOT, "oracle", "_build", "libpf_oracle.so")
    srcs = [os.path.join(ROOT, "oracle", "pf_oracle.cpp"), os.path.join(ROOT, "oracle", "philox.h"),
            os.path.join(ROOT, "include", "pointfoot_hip.h")]
    _build(["make", "-s", "-C", "oracle"], out, srcs)
    lib = C.CDLL(out)
    lib.pfo_step.argtypes = [C.POINTER(_abi.PfModel), C.POINTER(_abi.PfConfig), C.POINTER(_abi.PfLayout),
                             C.POINTER(_abi.PfBuffers), C.c_void_p, C.c_int64, C.POINTER(_abi.PfInject), C.c_int]
    lib.pfo_energy.argtypes = [C.POINTER(_abi.PfModel), C.POINTER(_abi.PfConfig), C.POINTER(_abi.PfLayout),
                               C.c_void_p, C.POINTER(C.c_double)]
    lib.pfo_set_threads.argtypes = [C.c_int]
    return lib


def emu_build():
    """compile (never load) the host build of the step program"""
    out = os.path.join(ROOT, "tests", "emu", "_build", "libpf_emu.so")
    srcs = [os.path.join(ROOT, "tests", "emu", "pf_emu.cpp")] + [
        os.path.join(ROOT, "pointfoot_amd", "csrc", f) for f in ("pf_step_kernel.h", "pf_simt.h", "pf_layout.h", "pf_lump.h")] + [
        os.path.join(ROOT, "include", "pointfoot_hip.h")]
    _build(["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "tests/emu/pf_emu.cpp", "-o", out],
           out, srcs)
    return out


def emu_lib():
    out = emu_build()
    lib = C.CDLL(out)
    lib.pfe_step.argtypes = [C.POINTER(_abi.PfModel), C.POINTER(_abi.PfConfig), C.POINTER(_abi.PfLayout),
                             C.POINTER(_abi.PfBuffers), C.c_void_p, C.c_longlong, C.POINTER(_abi.PfInject), C.c_int, C.c_void_p]
    lib.pfe_layout.argtypes = [C.POINTER(_abi.PfModel), C.POINTER(_abi.PfConfig), C.POINTER(_abi.PfLayout)]
    lib.pfe_lump.argtypes = [C.POINTER(_abi.PfModel), C.POINTER(_abi.PfModel), C.c_void_p]
    lib.pfe_lump.restype = None
    return lib


def make_cfg(task, num_envs, control_type="P", add_noise=True, variant=None):
    """this repo's config instance for a test task name; `variant` / an "_allrew" / "_onlypos" suffix of a pointfoot task name
    applies tests/golden/variants.py (the same function the golden generator applies to the reference's config)"""
    import variants
    from pointfoot_amd.envs.pointfoot.mixed_terrain.pointfoot_rough_config import PointFootRoughCfg
    from pointfoot_amd.envs.pointfoot.flat.pointfoot_flat_config import PointFootFlatCfg
    if task.startswith("anymal"):
        from pointfoot_amd.envs.anymal_c.mixed_terrains.anymal_c_rough_config import AnymalCRoughCfg
        from pointfoot_amd.envs.anymal_c.flat.anymal_c_flat_config import AnymalCFlatCfg
        cfg = AnymalCRoughCfg() if task.endswith("rough") else AnymalCFlatCfg()
        cfg.env.num_envs = num_envs
        cfg.control.control_type = control_type
        cfg.noise.add_noise = add_noise
        return variants.apply(cfg, variant)
    cfg = PointFootRoughCfg() if task.startswith("rough") else PointFootFlatCfg()
    variants.apply(cfg, variant if variant is not None else task)
    if task == "rough_hactor":      # point_foot.py:361: height measurements in the actor's observations too
        cfg.terrain.measure_heights_actor = True
        cfg.env.num_propriceptive_obs = 27 + len(cfg.terrain.measured_points_x) * len(cfg.terrain.measured_points_y)
    cfg.env.num_envs = num_envs
    cfg.control.control_type = control_type
    cfg.noise.add_noise = add_noise
    return cfg


class HostSim:
    """Numpy-backed PfBuffers + state record, steppable by the oracle or the emu build."""

    def __init__(self, cfg, n, hf=None, terrain_origins=None, mean_base_mass=None, seed=1, actuator=None):
        self.cfg = cfg
        self.n = n
        sp = SimParams(dt=cfg.sim.dt)
        self.model, self.info = _native.build_model(cfg)
        kind = _abi.PF_KIND_LEGGED_ROBOT if hasattr(cfg.env, "num_observations") else _abi.PF_KIND_POINTFOOT
        self.c, self.reward_names = _native.build_config(
            cfg, self.model, self.info, sp, n, hf_shape=None if hf is None else hf.shape,
            mean_base_mass=mean_base_mass, seed=seed, kind=kind)
        self.actuator = None if actuator is None else np.ascontiguousarray(actuator, np.float32)
        if self.actuator is None:
            self.c.use_actuator_net = 0
        # the record layout comes from the HIP library's own pf_layout (a host-only entry point: no GPU needed), so a GPU test
        # process never maps the CPU build of the kernel source; tests/test_host_logic.py cross-checks pfe_layout against it
        self.L = _abi.PfLayout()
        _abi.check(_abi.load().pf_layout(C.byref(self.model), C.byref(self.c), C.byref(self.L)))
        L = self.L
        self.state = np.zeros((n, L.rec_floats), np.float32)
        self.state[:, L.root + 6] = 1.0
        self.episode_length = np.zeros(n, np.int64)
        self.terrain_level = np.zeros(n, np.int32)
        self.terrain_type = np.zeros(n, np.int32)
        self.reset = np.zeros(n, np.uint8)
        self.time_out = np.zeros(n, np.uint8)
        self.terminate = np.zeros(n, np.uint8)
        self.obs = np.zeros((n, self.c.num_obs), np.float32)
        self.priv = np.zeros((n, max(self.c.num_priv_obs, 1)), np.float32)
        self.rew = np.zeros(n, np.float32)
        self.heights = np.zeros((n, max(self.c.num_height_points, 1)), np.float32)
        self.hf = None if hf is None else np.ascontiguousarray(hf, np.int16)
        self.terrain_origins = None if terrain_origins is None else np.ascontiguousarray(terrain_origins, np.float32)
        self.stats = np.zeros(_abi.PF_STATS_FLOATS, np.float32)
        self.contact_set = np.zeros((n, self.c.decimation, _abi.PF_CSET_WORDS), np.uint64)  # diagnostic: contact candidates / joint-limit rows / cells per substep
        self.joint_friction = None                                          # (n, nd) float32 or None
        self.bufs = self.make_bufs()

    def make_bufs(self):
        p = _native.ptr
        b = _abi.PfBuffers()
        b.d_state, b.d_episode_length = p(self.state), p(self.episode_length)
        b.d_terrain_level, b.d_terrain_type = p(self.terrain_level), p(self.terrain_type)
        b.d_reset, b.d_time_out, b.d_terminate = p(self.reset), p(self.time_out), p(self.terminate)
        b.d_obs, b.d_priv, b.d_rew = p(self.obs), p(self.priv), p(self.rew)
        b.d_measured_heights = p(self.heights)
        b.d_height_samples = p(self.hf)
        b.d_terrain_origins = p(self.terrain_origins)
        b.d_actuator_params = p(self.actuator)
        b.d_reset_stats = p(self.stats)
        b.d_contact_set = p(self.contact_set)
        b.d_joint_friction = p(self.joint_friction)
        return b

    def f(self, name, width=None):
        """view of a record field"""
        o = getattr(self.L, name)
        return self.state[:, o:o + width] if width else self.state[:, o]

    def clone(self):
        import copy
        other = copy.copy(self)
        for k in ("state", "episode_length", "terrain_level", "terrain_type", "reset", "time_out", "terminate", "obs",
                  "priv", "rew", "heights", "stats", "contact_set"):
            setattr(other, k, getattr(self, k).copy())
        other.bufs = other.make_bufs()
        return other

    def step(self, backend, actions, counter, inject=None, physics_bits=32, mode=0, reset_mask=None):
        actions = np.ascontiguousarray(actions, np.float32)
        inj = inject if inject is not None else None
        if backend == "oracle":
            assert mode == 0
            rc = oracle_lib().pfo_step(C.byref(self.model), C.byref(self.c), C.byref(self.L), C.byref(self.bufs),
                                       actions.ctypes.data, counter, inj, physics_bits)
        elif backend == "emu":
            rc = emu_lib().pfe_step(C.byref(self.model), C.byref(self.c), C.byref(self.L), C.byref(self.bufs),
                                    actions.ctypes.data, counter, inj, mode,
                                    None if reset_mask is None else np.ascontiguousarray(reset_mask, np.uint8).ctypes.data)
        else:
            raise ValueError(backend)
        assert rc == 0


def load_golden(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def sim_from_golden(g, task, control_type="P", add_noise=True):
    """HostSim initialised to the golden file's `init_*` snapshot."""
    import synth
    n = g["init_root"].shape[0]
    cfg = make_cfg(task, n, control_type, add_noise)
    rough = task.startswith("rough")
    hf = synth.synth_heightfield() if rough else None
    to = g["terrain_origins"] if rough else None
    sim = HostSim(cfg, n, hf=hf, terrain_origins=to, mean_base_mass=float(g["base_mass"].mean(dtype=np.float32)))
    L = sim.L
    nd, nf = 6, 2
    sim.f("root", 13)[:] = g["init_root"]
    sim.f("dof_pos", nd)[:] = g["init_dof"][:, :, 0]
    sim.f("dof_vel", nd)[:] = g["init_dof"][:, :, 1]
    sim.f("commands", 4)[:, :g["init_commands"].shape[1]] = g["init_commands"]
    sim.f("last_actions", nd)[:] = g["init_last_actions"]
    sim.f("last_dof_vel", nd)[:] = g["init_last_dof_vel"]
    for name in ("feet_air_time", "last_feet_air_time", "contact_filt", "first_contact", "last_contacts", "feet_height",
                 "last_max_feet_height"):
        sim.f(name, nf)[:] = g["init_" + name].astype(np.float32)
    sim.f("cur_max_feet_height", nf)[:] = g["init_current_max_feet_height"]
    R = g["init_episode_sums"].shape[0]          # reward terms (+ the termination sum, kept last, when its scale is non-zero)
    sim.f("episode_sums", R)[:] = g["init_episode_sums"].T
    sim.f("env_origin", 3)[:] = g["init_env_origins"]
    sim.episode_length[:] = g["init_episode_length"]
    if rough:
        sim.terrain_level[:] = g["init_terrain_levels"]
        sim.terrain_type[:] = g["terrain_types"]
    return sim


def inject_from_golden(g, t, sim, keep):
    """PfInject for golden step t (arrays are appended to `keep` to stay alive)."""
    L = sim.L
    n = sim.n
    p = f"s{t}"
    rnd = np.zeros((n, L.rand_floats), np.float32)
    rnd[:, L.r_obs:L.r_obs + g[p + "_r_obs"].shape[1]] = g[p + "_r_obs"]
    rp = g[p + "_r_priv"]
    if rp.shape[1]:
        rnd[:, L.r_priv:L.r_priv + rp.shape[1]] = rp
    rnd[:, L.r_dof:L.r_dof + 6] = g[p + "_r_dof"]
    rnd[:, L.r_xy:L.r_xy + 2] = g[p + "_r_xy"]
    rnd[:, L.r_vel:L.r_vel + 6] = g[p + "_r_vel"]
    rnd[:, L.r_level] = np.maximum(g[p + "_r_level"], 0)
    rnd[:, L.r_push:L.r_push + 3] = g[p + "_r_push"]
    arrs = dict(d_rand=rnd, d_dof_pre=np.ascontiguousarray(g[p + "_dof_pre"]),
                d_dof_post=np.ascontiguousarray(g[p + "_dof_post"]), d_root=np.ascontiguousarray(g[p + "_sim_root"]),
                d_contact=np.ascontiguousarray(g[p + "_sim_contact"]),
                d_feet_pos=np.ascontiguousarray(g[p + "_sim_feet_pos"]))
    inj = _abi.PfInject()
    for k, a in arrs.items():
        setattr(inj, k, _native.ptr(a))
    keep.append(arrs)
    return inj, arrs


class HipMirror:
    """Runs a HostSim's buffers through the real HIP library: upload -> pf_step -> download."""
    FIELDS = ("state", "episode_length", "terrain_level", "terrain_type", "reset", "time_out", "terminate", "obs",
              "priv", "rew", "heights", "stats", "contact_set")

    def __init__(self, sim, device="cuda:0"):
        import torch
        self.torch = torch
        self.sim = sim
        self.dev = device
        self.lib = _abi.load()
        self.t = {}
        for k in self.FIELDS:
            a = getattr(sim, k)
            self.t[k] = torch.from_numpy((a.view(np.int64) if a.dtype == np.uint64 else a).copy()).to(device)
        self.t["jf"] = None if sim.joint_friction is None else torch.from_numpy(sim.joint_friction).to(device)
        self.t["hf"] = None if sim.hf is None else torch.from_numpy(sim.hf).to(device)
        self.t["to"] = None if sim.terrain_origins is None else torch.from_numpy(sim.terrain_origins).to(device)
        self.h = self.lib.pf_create(C.byref(sim.model), C.byref(sim.c), 0)
        assert self.h, self.lib.pf_last_error()
        p = _native.ptr
        b = _abi.PfBuffers()
        t = self.t
        b.d_state, b.d_episode_length = p(t["state"]), p(t["episode_length"])
        b.d_terrain_level, b.d_terrain_type = p(t["terrain_level"]), p(t["terrain_type"])
        b.d_reset, b.d_time_out, b.d_terminate = p(t["reset"]), p(t["time_out"]), p(t["terminate"])
        b.d_obs, b.d_priv, b.d_rew = p(t["obs"]), p(t["priv"]), p(t["rew"])
        b.d_measured_heights = p(t["heights"])
        b.d_height_samples, b.d_terrain_origins = p(t["hf"]), p(t["to"])
        t["act"] = None if sim.actuator is None else torch.from_numpy(sim.actuator).to(device)
        b.d_actuator_params = p(t["act"])
        b.d_reset_stats = p(t["stats"])
        b.d_contact_set = p(t["contact_set"])
        b.d_joint_friction = p(t["jf"])
        self.b = b
        _abi.check(self.lib.pf_bind_buffers(self.h, C.byref(b)), self.lib)
        Lh = _abi.PfLayout()
        _abi.check(self.lib.pf_layout(C.byref(sim.model), C.byref(sim.c), C.byref(Lh)), self.lib)
        assert bytes(Lh) == bytes(sim.L), "pf_layout (HIP library) disagrees with the test layout"

    def upload(self):
        for k in self.FIELDS:
            a = getattr(self.sim, k)
            self.t[k].copy_(self.torch.from_numpy(a.view(np.int64) if a.dtype == np.uint64 else a))

    def download(self):
        self.torch.cuda.synchronize()
        for k in self.FIELDS:
            a = getattr(self.sim, k)
            a[...] = self.t[k].cpu().numpy().view(a.dtype)

    def step(self, actions, counter, inject_arrays=None, mode=0, reset_mask=None):
        torch = self.torch
        a = torch.from_numpy(np.ascontiguousarray(actions, np.float32)).to(self.dev)
        inj = None
        keep = []
        if inject_arrays is not None:
            inj = _abi.PfInject()
            for k, arr in inject_arrays.items():
                tt = torch.from_numpy(arr).to(self.dev)
                keep.append(tt)
                setattr(inj, k, _native.ptr(tt))
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        mask_t = None if reset_mask is None else torch.from_numpy(np.ascontiguousarray(reset_mask, np.uint8)).to(self.dev)
        fn = {0: lambda: self.lib.pf_step(self.h, a.data_ptr(), counter, inj, stream),
              1: lambda: (self.lib.pf_reset_all(self.h, counter, inj, stream) if reset_mask is None else
                          self.lib.pf_reset_idx(self.h, mask_t.data_ptr(), counter, inj, stream)),
              2: lambda: self.lib.pf_compute_observations(self.h, counter, inj, stream)}[mode]
        _abi.check(fn(), self.lib)
        self.download()

    def close(self):
        if self.h:
            self.lib.pf_destroy(self.h)
            self.h = None


class EmuMirror:
    """Debug aid: the HipMirror interface on top of the host build of the step program, so the `-m gpu` env tests can be
    rehearsed on a machine without a GPU (`PF_TEST_EMU=1 pytest -m gpu tests/test_gpu_env.py`).  Proves nothing about
    the GPU; the round's GPU tests run with the real library."""
    FIELDS = HipMirror.FIELDS

    def __init__(self, sim, device=None):
        self.sim = sim

    def upload(self):
        pass

    def download(self):
        pass

    def step(self, actions, counter, inject_arrays=None, mode=0, reset_mask=None):
        inj = None
        if inject_arrays is not None:
            inj = _abi.PfInject()
            for k, arr in inject_arrays.items():
                setattr(inj, k, _native.ptr(arr))
        self.sim.step("emu", actions, counter, inject=inj, mode=mode, reset_mask=reset_mask)

    def close(self):
        pass


if os.environ.get("PF_TEST_EMU") == "1":
    HipMirror = EmuMirror


def record_parity(test, **measured):
    """Leave what a `-m gpu` parity test MEASURED (flipped-decision fractions, allowance counts, worst GEMM error / bound ...)
    where the GPU run's caller can see it: one JSON object per test in gpurun_out/parity_fractions.json (pytest -q swallows
    prints).  The end-of-round copy is committed as profiles/r<N>_parity_fractions.json.  Host-emulation rehearsals do not write."""
    import json
    if os.environ.get("PF_TEST_EMU") == "1":
        return
    path = os.environ.get("PF_PARITY_JSON", os.path.join(ROOT, "gpurun_out", "parity_fractions.json"))
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        try:
            with open(path) as f:
                data = json.load(f)
        except (OSError, ValueError):
            data = {}
        clean = {}
        for k, v in measured.items():
            clean[k] = v.item() if hasattr(v, "item") else v
        data[test] = clean
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
