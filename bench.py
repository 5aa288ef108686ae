#!/usr/bin/env python3
"""bench.py — headline benchmark: whole-node env-steps/s of the PPO training loop on `pointfoot_rough`,
4096 envs per GPU (BASELINE.json metric).

One "step" = one PPO iteration on every rank: 24 policy steps of the fused env kernel driven by the live actor
(rollout + transition storage), GAE, and the full update (5 epochs x 4 minibatches: forward, loss, backward,
gradient all-reduce when N>1, grad-norm clip, Adam, adaptive-KL LR).  Inputs are synthetic (seeded terrain, random-init
policy) and resident in HBM before the timed region.  value = N_gpus * 4096 * 24 * K / max-over-ranks wall time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no torchrun environment starts the N ranks ITSELF (`self_launch`): the parent never
imports torch, never touches the GPU, starts `python -m torch.distributed.run` as a child process, relays rank 0's one JSON line
and exits with the child's code.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--num_envs", type=int, default=4096)
    p.add_argument("--task", type=str, default="pointfoot_rough")
    p.add_argument("--no_cpu_baseline", action="store_true")
    p.add_argument("--no_alt", action="store_true", help="skip the second timed loop on the fp32 MFMA kernels")
    p.add_argument("--gemm", choices=["auto", "fp32", "bf16x6"], default="auto",
                   help="dense-layer arithmetic: fp32 MFMA, fp32 by exact 3-way bf16 split on the bf16 MFMA, or by network width (default)")
    p.add_argument("--backend", type=str, default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    p.add_argument("--single_device", action="store_true", help="rehearsal: every rank uses cuda:0 (with --backend gloo)")
    p.add_argument("--self_launch", action="store_true",
                   help="start the ranks through torch.distributed.run from a parent that never touches the GPU even when --gpus 1 "
                        "(always done for --gpus N > 1 outside torchrun)")
    p.add_argument("--master_port", type=int, default=0, help="rendezvous port of the self-launched job (0: pick a free one)")
    return p.parse_args(argv)


def under_torchrun(env=None):
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def needs_self_launch(a, env=None):
    """True in the parent of a multi-rank run: --gpus N > 1 (or --self_launch) and no torchrun environment yet."""
    return (a.gpus > 1 or a.self_launch) and not under_torchrun(env)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch_command(a, argv, port):
    """The child a parent starts: one process per GPU of ONE node over RCCL -- exactly the command line the driver uses."""
    child_args = [x for x in argv if x != "--self_launch"]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_args


def self_launch(a, argv):
    """Parent of an N-rank run.  Nothing here imports torch or makes a HIP call: the ranks are fresh child processes (never an
    exec of a process that has initialised the GPU).  stdout of the job = the ONE JSON line rank 0 prints; everything else the
    ranks write (stderr, banners) passes through on stderr."""
    assert "torch" not in sys.modules, "bench.py parent imported torch before starting its ranks"
    port = a.master_port or free_port()
    cmd = self_launch_command(a, argv, port)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    print("[bench] starting ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True, cwd=os.getcwd())
    line = None
    for ln in proc.stdout:
        s = ln.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        elif s:
            print(s, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("[bench] the ranks exited 0 without a result line", file=sys.stderr, flush=True)
        rc = 1
    return rc


if __name__ == "__main__" and needs_self_launch(parse()):
    sys.exit(self_launch(parse(), sys.argv[1:]))

import numpy as np
import torch
import torch.distributed as dist

NUM_ENVS = 4096
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 matrix peak
MFMA_BF16_PEAK_TFLOPS = 2500.0 # MI355X_MICROARCH.md: dense bf16 MFMA peak
STEP_BYTES_PER_ENV = 2072      # algorithmic bytes per env-step, pointfoot_rough
PPO_MFLOP_PER_ENV_STEP = 13.4  # 0.84 rollout + 12.5 update
# HBM bytes per step-kernel launch come from the committed rocprofv3 PMC summary of THIS kernel (tools/pmc_traffic.py
# writes it from separate --pmc passes: the L2's memory-side read / write requests by size -- the exact form of the guide's gfx950
# doubling rule, profiles/r4_fetch_size_gather_calibration.txt); used only when kernel, task and env count match.
TRAFFIC_PROFILES = [os.path.join(ROOT, "profiles", f) for f in
                    ("r4_step_kernel_hbm_pmc_policy_step.json",      # the launch WITH the policy's tail (pf_step_policy: "policy_tail": true)
                     "r4_step_kernel_hbm_pmc.json", "r4_step_kernel_hbm_pmc_anymal_c_rough.json")]      # profiles of the kernels as they are NOW
# (older summaries -- r3_step_kernel_hbm_pmc*.json, r2_... -- describe kernels that no longer exist: single-wavefront PointFoot instance,
# whole-record loads; they are not offered as this build's traffic)


def step_traffic(kernel_prefix, task, num_envs, tail=False):
    for path in TRAFFIC_PROFILES:      # a file is used only for the kernel instance, task, env count and kind of launch (with / without the policy's tail) it was measured on
        try:
            with open(path) as f:
                p = json.load(f)
            if kernel_prefix in p["kernel"].replace(" ", "") and p["task"] == task and int(p["num_envs"]) == num_envs and bool(p.get("policy_tail", False)) == bool(tail):
                return float(p["traffic_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def make(task, num_envs, device):
    import contextlib
    from pointfoot_amd.envs import task_registry  # registers tasks
    from pointfoot_amd.utils import get_args
    args = get_args(["--task", task, "--headless", "--num_envs", str(num_envs), "--sim_device", device, "--rl_device", device])
    with contextlib.redirect_stdout(sys.stderr):      # the registry / ActorCritic print like the reference does; stdout carries the JSON line only
        env, env_cfg = task_registry.make_env(name=task, args=args)
        runner, train_cfg = task_registry.make_alg_runner(env=env, name=task, args=args, log_root=None)
    return env, runner, train_cfg


def cpu_baseline(task, num_envs, threads):
    """The CPU restatement of the same iteration, timed on this host: oracle C++ env step (OpenMP) + torch-CPU PPO
    update with the reference hyper-parameters.  Bounded sample: ONE iteration (num_envs x 24 env-steps)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import common
    import ppo_ref
    from pointfoot_amd.utils.terrain import Terrain
    cfg = common.make_cfg("rough" if "rough" in task else "flat", num_envs)
    rough = cfg.terrain.mesh_type != "plane"
    np.random.seed(1)
    hf = to = None
    if rough:
        ter = Terrain(cfg.terrain, num_envs)
        hf, to = ter.heightsamples, ter.env_origins.astype(np.float32)
    sim = common.HostSim(cfg, num_envs, hf=hf, terrain_origins=to)
    rs = np.random.RandomState(0)
    if rough:
        sim.terrain_level[:] = rs.randint(0, 6, num_envs)
        sim.terrain_type[:] = np.arange(num_envs) // max(num_envs // 20, 1) % 20
        sim.f("env_origin", 3)[:] = to[sim.terrain_level, sim.terrain_type]
    sim.f("root", 13)[:, :3] = sim.f("env_origin", 3)
    sim.f("root", 13)[:, 2] += 0.62
    sim.f("friction")[:] = rs.uniform(0, 1.6, num_envs)
    common.oracle_lib().pfo_set_threads(threads)
    torch.set_num_threads(threads)
    T, O, P, A = 24, sim.c.num_obs, sim.c.num_priv_obs, 6
    hidden = [512, 256, 128] if rough else [128, 64, 32]
    ac = ppo_ref.RefActorCritic(O, P, A, hidden, hidden)
    opt = torch.optim.Adam(ac.parameters(), lr=1e-3)
    cfgd = dict(value_loss_coef=1.0, use_clipped_value_loss=True, clip_param=0.2, entropy_coef=0.01, num_learning_epochs=5,
                num_mini_batches=4, schedule="adaptive", desired_kl=0.01, max_grad_norm=1.0)
    t0 = time.time()
    S = {k: [] for k in ("obs", "critic_obs", "actions", "values", "old_log_prob", "old_mu", "old_sigma", "rewards", "dones")}
    obs, cobs = torch.from_numpy(sim.obs), torch.from_numpy(sim.priv[:, :P])
    for t in range(T):
        r = ppo_ref.act(ac, obs, cobs, torch.randn(num_envs, A))
        S["obs"].append(obs.clone()); S["critic_obs"].append(cobs.clone())
        sim.step("oracle", r["actions"].numpy(), t + 1)
        rew = ppo_ref.process_rewards(torch.from_numpy(sim.rew.copy()), r["values"], torch.from_numpy(sim.time_out.astype(bool)), 0.99)
        for k, v in (("actions", r["actions"]), ("values", r["values"][:, 0]), ("old_log_prob", r["log_prob"]), ("old_mu", r["mu"]),
                     ("old_sigma", r["sigma"]), ("rewards", rew), ("dones", torch.from_numpy(sim.reset.astype(np.float32)))):
            S[k].append(v)
    t_collect = time.time() - t0
    with torch.no_grad():
        lv = ac.critic(cobs)[:, 0]
    V, Rw, D = torch.stack(S["values"]), torch.stack(S["rewards"]), torch.stack(S["dones"])
    ret, adv, _ = ppo_ref.compute_returns(V, Rw, D, lv, 0.99, 0.95)
    store = dict(obs=torch.cat(S["obs"]), critic_obs=torch.cat(S["critic_obs"]), actions=torch.cat(S["actions"]), values=V.flatten(),
                 returns=ret.flatten(), advantages=adv.flatten(), old_log_prob=torch.cat(S["old_log_prob"]),
                 old_mu=torch.cat(S["old_mu"]), old_sigma=torch.cat(S["old_sigma"]))
    ppo_ref.update(ac, opt, store, torch.randperm(num_envs * T), cfgd, 1e-3)
    total = time.time() - t0
    torch.set_num_threads(max(1, min(usable_cpus(), 16)))
    return dict(value=num_envs * T / total, unit="env-steps/s", cores=threads, kind="port",
                sample=f"1 PPO iteration = {num_envs} envs x {T} steps (oracle C++ env step with OpenMP: {t_collect:.2f}s; "
                       f"torch-CPU PPO 5x4 minibatches: {total - t_collect:.2f}s); restatement CPU baseline, reference binary unavailable",
                host_cpus=os.cpu_count(), cpu_model=cpu_model())


class PowerSampler:
    """Socket power and shader clock WHILE the timed loop runs (its real duty cycle: ~10 ms of GEMM-bound update, ~4 ms of
    latency-bound rollout per iteration), from a side thread: the amdgpu hwmon / pp_dpm_sclk files of the card when readable (no
    process start, ~20 samples a second), else `rocm-smi --json` (a few samples a second).  Host-side only; nothing touches the
    GPU queue."""

    def __init__(self, period=0.05, pci_bus_id=None):
        import glob
        import threading
        self.period, self.samples, self._stop = period, [], False
        self.src = None
        devs = sorted(glob.glob("/sys/class/drm/card*/device"))
        if pci_bus_id is not None:
            # a GPU box shows this process ONE of the host's cards: take the hwmon files of the card on the same PCI bus
            mine = [d for d in devs if os.path.basename(os.path.realpath(d)).lower().split(":")[1:2] == [f"{pci_bus_id:02x}"]]
            devs = mine if mine else []
        for dev in devs:
            hw = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
            if hw and os.path.exists(os.path.join(dev, "pp_dpm_sclk")):
                pw = [f for f in ("power1_average", "power1_input") if os.path.exists(os.path.join(hw[0], f))]
                if pw:
                    self.src = ("sysfs", os.path.join(hw[0], pw[0]), os.path.join(dev, "pp_dpm_sclk"),
                                os.path.join(hw[0], "freq1_input") if os.path.exists(os.path.join(hw[0], "freq1_input")) else None,
                                os.path.join(hw[0], "power1_cap") if os.path.exists(os.path.join(hw[0], "power1_cap")) else None)
                    break
        if self.src is None:
            import shutil
            if shutil.which("rocm-smi"):
                self.src = ("rocm-smi",)
                self.period = max(period, 0.15)
        self._thread = threading.Thread(target=self._run, daemon=True)

    def _read(self):
        if self.src[0] == "sysfs":
            _, pw, dpm, freq, _cap = self.src
            with open(pw) as f:
                watts = int(f.read().strip()) / 1e6
            mhz = None
            if freq:
                with open(freq) as f:
                    mhz = int(f.read().strip()) / 1e6
            else:
                with open(dpm) as f:
                    for ln in f:
                        if "*" in ln:
                            mhz = float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            return watts, mhz
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        card = next(iter(json.loads(out).values()))
        watts = next((float(v) for k, v in card.items() if "Power" in k and "W" in k), None)
        sclk = next((v for k, v in card.items() if k.startswith("sclk")), None)
        mhz = float(str(sclk).lower().replace("mhz", "").strip("() ")) if sclk is not None else None
        return watts, mhz

    def _run(self):
        while not self._stop:
            try:
                w, m = self._read()
                self.samples.append((time.perf_counter(), w, m))
            except Exception:
                pass
            time.sleep(self.period)

    def one(self):
        try:
            w, m = self._read()
            return {"socket_w": w, "sclk_mhz": m}
        except Exception:
            return None

    def start(self):
        if self.src is not None:
            self._thread.start()

    def stop(self):
        self._stop = True
        if self._thread.is_alive():
            self._thread.join(timeout=2)

    def summary(self, t0, t1):
        inside = [(w, m) for t, w, m in self.samples if t0 <= t <= t1]
        st = lambda v: ({"min": float(min(v)), "median": float(np.median(v)), "max": float(max(v))} if v else None)
        cap = None
        if self.src and self.src[0] == "sysfs" and self.src[4]:
            try:
                with open(self.src[4]) as f:
                    cap = int(f.read().strip()) / 1e6
            except Exception:
                pass
        return {"source": (self.src[0] + (":" + self.src[1].split("/device/")[0].rsplit("/", 1)[-1] if self.src[0] == "sysfs" else "")) if self.src else None,
                "samples": len(inside), "power_cap_w": cap,
                "socket_w": st([w for w, _ in inside if w is not None]), "sclk_mhz": st([m for _, m in inside if m is not None])}


def usable_cpus():
    """cores this process may actually use: scheduler affinity capped by the cgroup CPU quota (a GPU box hands each job a
    share of a much larger host; os.cpu_count() reports the whole host)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(float(parts[0]) / float(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, q // int(g.read().split()[0])))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    a = parse()
    if a.gemm != "auto":
        os.environ["PF_PPO_GEMM"] = "1" if a.gemm == "fp32" else "2"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.single_device:
        local = 0
    ndev = torch.cuda.device_count()
    if local >= ndev:      # e.g. --gpus 8 on a node with fewer GPUs: say so instead of a HIP "invalid device ordinal" from deep inside
        sys.exit(f"bench.py: rank {rank} (LOCAL_RANK {local}) needs cuda:{local}, but this node shows {ndev} GPU(s); --gpus {a.gpus} "
                 f"is one process per GPU of ONE node (--backend gloo --single_device rehearses N > 1 on one GPU)")
    # PF_DP_SINGLE_RANK_COLLECTIVES=1: a process group of ONE rank with every collective of the N>1 path issued anyway (dp.collectives_on):
    # the rehearsal of the multi-GPU code path through RCCL on a one-GPU box
    rehearse1 = world == 1 and os.environ.get("PF_DP_SINGLE_RANK_COLLECTIVES", "0") == "1"
    if rehearse1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    dist_on = world > 1 or rehearse1
    if dist_on:
        torch.cuda.set_device(local)
        # the collective libraries print connection banners to fd 1 from C++ (gloo: "[Gloo] Rank 0 is connected to ..."): stdout
        # carries the one JSON line only, so fd 1 points at stderr while the process group comes up
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            from pointfoot_amd import dp
            dp.init_process_group(a.backend, local)       # nccl: bound to cuda:<local> (device_id)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    env, runner, train_cfg = make(a.task, a.num_envs, device)
    T = runner.num_steps_per_env
    runner.env.episode_length_buf = torch.randint_like(runner.env.episode_length_buf, high=int(runner.env.max_episode_length))
    obs = env.get_observations()
    cobs = env.get_privileged_observations()
    cobs = cobs if cobs is not None else obs

    ev_pairs = []           # (start, stop) around every fused env-step launch, on the launch stream
    up_pairs = []
    orig_step = env.step

    # Every EV_EVERY-th env-step launch of the timed region is bracketed by HIP events (3 of a rollout's 24): an event pair costs the
    # stream ~13 us of bubbles (tools/halfbatch_probe.py: the same rollout takes 3.44 ms without them, 3.76 ms with a pair around
    # every step) -- instrumentation that would otherwise sit inside the metric's own timed region.
    EV_EVERY = 8
    step_no = [0]

    def timed(fn):
        def wrapped(arg):
            step_no[0] += 1
            if step_no[0] % EV_EVERY:
                return fn(arg)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(arg)
            e.record()
            ev_pairs.append((s, e))
            return out
        return wrapped

    timed_step = timed(orig_step)
    orig_step_policy = getattr(env, "step_policy", None)       # PF_FUSED_POLICY_STEP=1: the runner steps the env through this one
    timed_step_policy = timed(orig_step_policy) if orig_step_policy else None

    it_marks = []           # events at the head of every timed iteration (+ one at the end): per-iteration durations -> medians

    def iteration(obs, cobs, record):
        env.step = timed_step if record else orig_step
        if orig_step_policy:
            env.step_policy = timed_step_policy if record else orig_step_policy
        if record:
            m = torch.cuda.Event(enable_timing=True)
            m.record()
            it_marks.append(m)
        obs, cobs, _ = runner.rollout(obs, cobs)
        if record:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
        runner.alg.update()
        if record:
            e.record()
            up_pairs.append((s, e))
        return obs, cobs

    sampler = PowerSampler(pci_bus_id=getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None)) if rank == 0 else None
    power_idle = sampler.one() if sampler and sampler.src else None
    for _ in range(a.warmup):
        obs, cobs = iteration(obs, cobs, False)
    torch.cuda.synchronize()
    if sampler:
        sampler.start()
    if rank == 0:
        print(f"[bench] warm-up done ({a.warmup} iterations)", file=sys.stderr, flush=True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        obs, cobs = iteration(obs, cobs, True)
    end_mark = torch.cuda.Event(enable_timing=True)
    end_mark.record()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t_end = time.perf_counter()
    dt = t_end - t0
    power = None
    if sampler:
        sampler.stop()
        power = sampler.summary(t0, t_end)
        power["before_warmup"] = power_idle
        if power["samples"] < 4 and sampler.src:
            # a timed region shorter than a few sampling periods: run the same alternating loop (untimed) until there are samples
            sampler2 = PowerSampler(pci_bus_id=getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None)); sampler2.start()
            t2 = time.perf_counter()
            while time.perf_counter() - t2 < 1.5:
                obs, cobs = iteration(obs, cobs, False)
                torch.cuda.synchronize()
            sampler2.stop()
            power = sampler2.summary(t2, time.perf_counter())
            power["before_warmup"] = power_idle
            power["note"] = "sampled over 1.5 s of the same alternating rollout / update loop run again after the timed region (the timed region was shorter than the sampling needs)"
        else:
            power["note"] = "sampled inside the timed region (rollout and update alternating as timed)"
    if dist_on:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_gpus = world
    env_steps = n_gpus * a.num_envs * T * a.steps
    value = env_steps / dt
    if rank == 0:
        print(f"[bench] timed region done: {a.steps} iterations in {dt:.3f} s = {value:.0f} env-steps/s", file=sys.stderr, flush=True)
    step_ms = float(np.mean([s.elapsed_time(e) for s, e in ev_pairs]))            # one fused env-step launch
    # the policy's tail inside the same launch (OnPolicyRunner.rollout: PF_FUSED_POLICY_STEP, default on): algorithmic bytes per env-step =
    # last hidden rows of both networks + the observation rows read and written once + action / mean / sigma rows + value + log-prob
    fused_step = os.environ.get("PF_FUSED_POLICY_STEP", "1") == "1" and hasattr(env, "step_policy")
    ac_ = runner.alg.actor_critic
    tail_bytes = 4 * (ac_.actor_dims[-2] + ac_.critic_dims[-2] + 2 * (env.num_obs + (env.num_privileged_obs or env.num_obs)) + 3 * env.num_actions + 2)
    upd_ms = float(np.mean([s.elapsed_time(e) for s, e in up_pairs]))
    # medians over the timed iterations, and the collection-only / learn-only rates.  Device-side durations (HIP
    # events on the launch stream): iteration i = mark i -> mark i+1, collection = mark i -> update start, learn = the update.
    marks = it_marks + [end_mark]
    it_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(len(it_marks))]
    col_ms = [it_marks[i].elapsed_time(up_pairs[i][0]) for i in range(len(it_marks))]
    upd_all = [s.elapsed_time(e) for s, e in up_pairs]
    med = lambda v: float(np.median(v))
    per_iter_steps = a.num_envs * T
    finite = bool(torch.isfinite(runner.alg.params).all().item()) and bool(torch.isfinite(env.rew_buf).all().item())
    # the same loop once more with the dense layers on the plain fp32 MFMA kernels (v_mfma_f32_32x32x2_f32), reported beside
    # `value` so that the two arithmetic paths can be compared from one run (single GPU, default --gemm only)
    alt = None
    ac0 = runner.alg.actor_critic
    if world == 1 and a.gemm == "auto" and not a.no_alt and os.environ.get("PF_PPO_GEMM") is None and \
            max(max(ac0.actor_dims[1:-1]), max(ac0.critic_dims[1:-1])) >= 256:
        import contextlib
        from pointfoot_amd.envs import task_registry
        from pointfoot_amd.utils import get_args
        os.environ["PF_PPO_GEMM"] = "1"
        try:
            args2 = get_args(["--task", a.task, "--headless", "--num_envs", str(a.num_envs), "--sim_device", device, "--rl_device", device])
            with contextlib.redirect_stdout(sys.stderr):
                runner2, _ = task_registry.make_alg_runner(env=env, name=a.task, args=args2, log_root=None)
            env.step = orig_step
            if orig_step_policy:
                env.step_policy = orig_step_policy
            o2, c2 = env.get_observations(), env.get_privileged_observations()
            c2 = c2 if c2 is not None else o2
            for _ in range(max(2, a.warmup // 2)):
                o2, c2, _ = runner2.rollout(o2, c2); runner2.alg.update()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                o2, c2, _ = runner2.rollout(o2, c2); runner2.alg.update()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t1
            alt = {"gemm": "fp32 MFMA (v_mfma_f32_32x32x2_f32)", "value": a.num_envs * T * a.steps / dt2, "ms_per_step": dt2 / a.steps * 1e3}
        finally:
            del os.environ["PF_PPO_GEMM"]
    lockstep = None
    dp_backend = None
    if dist_on:
        from pointfoot_amd import dp
        dp_backend = dp.backend_name()
        # every rank must hold the same policy and have taken the same learning-rate decisions: compare a checksum of the
        # flat parameter buffer (as exact integers: the float32 bit patterns summed in int64) and the LR scalar across ranks
        alg = runner.alg
        chk = torch.stack([alg.params.view(torch.int32).to(torch.int64).sum(), alg.scalars[0:1].view(torch.int32).to(torch.int64)[0]])
        allchk = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allchk, chk)
        lockstep = all(bool(torch.equal(c, allchk[0])) for c in allchk)
    if rank == 0:
        rough = "rough" in a.task
        anymal = a.task.startswith("anymal")
        ac = runner.alg.actor_critic
        fwd_macs = sum(x * y for x, y in zip(ac.actor_dims[:-1], ac.actor_dims[1:])) + sum(x * y for x, y in zip(ac.critic_dims[:-1], ac.critic_dims[1:]))
        upd_flop_per_env_step = 2.0 * fwd_macs * 3 * runner.alg.num_learning_epochs        # fwd + 2x bwd per epoch (SURVEY §8d accounting)
        if anymal:
            bytes_per_env = 5900 if rough else 3900          # anymal_c_rough ~5.9 KB/env-step
        else:
            bytes_per_env = STEP_BYTES_PER_ENV if rough else 850
        ach_gbs = bytes_per_env * a.num_envs / (step_ms * 1e-3) / 1e9
        upd_flop = upd_flop_per_env_step * a.num_envs * T
        ach_tf = upd_flop / (upd_ms * 1e-3) / 1e12
        headline = a.task == "pointfoot_rough" and a.num_envs == NUM_ENVS
        dims = lambda d: "-".join(str(v) for v in d)
        kern = ("pf_step_kernel<17,12,4,LeggedRobot> (one wavefront per robot, four robots per workgroup)" if anymal
                else "pf_step_kernel<9,6,2,PointFoot,DUAL> (two wavefronts per pair of robots: dynamics beside contacts)")
        widest = max(max(ac.actor_dims[1:-1]), max(ac.critic_dims[1:-1]))
        env_gemm = os.environ.get("PF_PPO_GEMM")           # set by --gemm; unset: the library picks by width (ppo_create)
        x6 = env_gemm == "2" or (env_gemm is None and widest >= 256)
        gemm_name = ("pf_gemm3_* (fp32 by exact three-way bf16 operand split on v_mfma_f32_32x32x16_bf16, fp32 accumulation)" if x6
                     else "pf_gemm_* (v_mfma_f32_32x32x2_f32)")
        traffic, traffic_src = step_traffic("pf_step_kernel<17" if anymal else "pf_step_kernel<9", a.task, a.num_envs, tail=fused_step)
        out = {
            "metric": "env-steps/sec (whole node), pointfoot_rough 4096 envs/GPU" if headline else f"env-steps/sec (whole node), {a.task} {a.num_envs} envs/GPU",
            "value": value, "unit": "env-steps/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.task}: {a.num_envs} envs/GPU x {T} policy steps/iter (4 physics substeps each), "
                                   + ("curriculum height field 1300x2100 int16, " + (f"{env._cfg_c.num_height_points}-pt height scan, ") if rough else "plane, ")
                                   + f"DR + noise + pushes; PPO {runner.alg.num_learning_epochs} epochs x {runner.alg.num_mini_batches} "
                                   f"minibatches of {a.num_envs * T // runner.alg.num_mini_batches}, actor {dims(ac.actor_dims)} / critic {dims(ac.critic_dims)} (ELU), fp32"
                                   + (" (dense layers: every fp32 operand split exactly into 3 bf16 terms, 6 of the 9 partial products on the bf16 MFMA, fp32 accumulation; dropped terms < 2^-23 relative)" if x6 else " (fp32 MFMA)")
                                   + (", ANYdrive actuator-network LSTM in the step kernel" if anymal else ""),
                       "parallelism": (f"dp{n_gpus} (env shards, gradient all-reduce over "
                                       + ("RCCL/xGMI" if dp_backend == "nccl" else f"{dp_backend} -- a rehearsal backend, NOT RCCL") + ")") if n_gpus > 1 else "single GPU",
                       "finite": finite},
            "ranks_in_lockstep": lockstep,
            # named from the process group itself (dp.backend_name: "nccl" = RCCL), not from the flag
            "collectives": (((f"{dp_backend}, group of ONE rank with every collective of the N>1 path issued (PF_DP_SINGLE_RANK_COLLECTIVES=1 rehearsal)" if rehearse1
                              else f"{dp_backend}, {world} ranks")
                             + ("; gradient all-reduce: two asynchronous ranges under the split backward pass" if os.environ.get("PF_PPO_OVERLAP_ALLREDUCE", "0") == "1"
                                else "; gradient all-reduce: one blocking call per minibatch")) if dist_on else None),
            "alt_fp32_mfma": alt,
            "roofline": {"kernel": kern + " (fused env step)", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": step_ms, "bytes_per_launch": bytes_per_env * a.num_envs,
                         "policy_tail": ({"in_this_launch": True, "bytes_per_launch": tail_bytes * a.num_envs,
                                          "achieved_with_tail": (bytes_per_env + tail_bytes) * a.num_envs / (step_ms * 1e-3) / 1e9,
                                          "note": "the launch also carries the policy's heads, the sampling and RolloutStorage.add_transitions (pf_step_policy: "
                                                  "last hidden rows read, observation rows copied, action / mean / sigma / value / log-prob stored), which used to be "
                                                  "a launch of their own (k_act_heads, 12 us); `achieved` / `frac` count the env step's SURVEY 8d bytes only, while "
                                                  "`traffic` (PMC) is that of the whole launch, tail included: compare it with bytes_per_launch + policy_tail.bytes_per_launch"} if fused_step else {"in_this_launch": False}),
                         "note": "VALU-issue- and latency-bound, not HBM-bound (SURVEY §8d honesty note; DESIGN §4.1); launch_ms from HIP events around every 8th launch, on the launch stream"},
            "roofline_ppo": {"kernel": gemm_name + " inside the PPO update", "bound": "mfma", "achieved": ach_tf,
                             "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / MFMA_F32_PEAK_TFLOPS,
                             "update_ms": upd_ms, "flop_per_update": upd_flop,
                             "mfma_issued": ({"dtype": "bf16", "achieved": 6.0 * ach_tf, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                              "frac": 6.0 * ach_tf / MFMA_BF16_PEAK_TFLOPS} if x6 else None),
                             "note": "algorithmic fp32 GEMM flops of one update / event-timed duration of the whole update phase, against the fp32 "
                                     "matrix peak; mfma_issued = the bf16 MFMA work actually issued (6 products per fp32 product) against the dense bf16 peak"},
            "phase_ms": {"env_step_launch": step_ms, "ppo_update": upd_ms, "iteration": dt / a.steps * 1e3,
                         "rollout": float(np.mean(col_ms)), "median": {"iteration": med(it_ms), "rollout": med(col_ms), "ppo_update": med(upd_all),
                                                                       "env_step_launch": med([s.elapsed_time(e) for s, e in ev_pairs])}},
            "power": power,
            "median_value": n_gpus * per_iter_steps / (med(it_ms) * 1e-3),
            "collection_rate": {"value": n_gpus * per_iter_steps / (med(col_ms) * 1e-3), "unit": "env-steps/s",
                                "note": "collection only: 24 x [policy forward + fused env step + transition store] + GAE, median iteration, this rank's device time"},
            "learn_rate": {"value": n_gpus * per_iter_steps / (med(upd_all) * 1e-3), "unit": "env-steps/s",
                           "note": "PPO update only (5 epochs x 4 minibatches), median iteration"},
        }
        if n_gpus == 1 and not a.no_cpu_baseline and not anymal:
            try:
                # once at the reference's physx.num_threads (10, pointfoot_rough_config.py:203) and once on
                # all host cores; one PPO iteration each (a bounded sample: a few seconds of CPU work per run)
                ncpu = usable_cpus()
                print(f"[bench] cpu baseline: 10 threads, then all {ncpu} usable cores (host reports {os.cpu_count()})", file=sys.stderr, flush=True)
                out["cpu_baseline"] = cpu_baseline(a.task, a.num_envs, min(10, ncpu))
                print(f"[bench] cpu baseline (10 threads): {out['cpu_baseline']['value']:.0f} env-steps/s", file=sys.stderr, flush=True)
                if ncpu > 10:
                    out["cpu_baseline"]["all_cores"] = cpu_baseline(a.task, a.num_envs, ncpu)
            except Exception as ex:      # the baseline is informational; never lose the GPU measurement over it
                out["cpu_baseline"] = {"value": None, "error": repr(ex)}
        print(json.dumps(out), flush=True)
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
