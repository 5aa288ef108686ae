"""CLI / config helpers (mirror of reference legged_gym/utils/helpers.py:41-190).

`get_args` reproduces the flags of helpers.py:152-178 *and* the ones Isaac Gym's
`gymutil.parse_arguments` adds (--sim_device, --pipeline, --graphics_device_id, --physx/--flex,
--num_threads, --subscenes, --slices) with plain argparse, so train.py / play.py command lines are
unchanged.  `parse_sim_params` returns a small dataclass in place of `gymapi.SimParams`.
"""
import argparse
import copy
import os
import random
from dataclasses import dataclass, field

import numpy as np
import torch

SIM_PHYSX, SIM_FLEX = 1, 0      # stand-ins for gymapi.SIM_PHYSX / SIM_FLEX


def class_to_dict(obj) -> dict:
    """helpers.py:41-56 — walks dir(), hence ALPHABETICAL key order (fixes the reward summation order)."""
    if not hasattr(obj, "__dict__"):
        return obj
    result = {}
    for key in dir(obj):
        if key.startswith("_"):
            continue
        val = getattr(obj, key)
        if isinstance(val, list):
            result[key] = [class_to_dict(item) for item in val]
        else:
            result[key] = class_to_dict(val)
    return result


def update_class_from_dict(obj, dict):
    for key, val in dict.items():
        attr = getattr(obj, key, None)
        if isinstance(attr, type):
            update_class_from_dict(attr, val)
        else:
            setattr(obj, key, val)
    return


def set_seed(seed):
    """helpers.py:67-77: numpy / random / device generators (the reference leaves torch's CPU generator alone)."""
    if seed == -1:
        seed = np.random.randint(0, 10000)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


@dataclass
class PhysxParams:
    use_gpu: bool = True
    num_subscenes: int = 0
    num_threads: int = 10
    solver_type: int = 1
    num_position_iterations: int = 4
    num_velocity_iterations: int = 0
    contact_offset: float = 0.01
    rest_offset: float = 0.0
    bounce_threshold_velocity: float = 0.5
    max_depenetration_velocity: float = 1.0
    max_gpu_contact_pairs: int = 2 ** 23
    default_buffer_size_multiplier: float = 5
    contact_collection: int = 2


@dataclass
class SimParams:
    """What the env reads from gymapi.SimParams: .dt, .use_gpu_pipeline, .physx.* (SURVEY §8b)."""
    dt: float = 0.005
    substeps: int = 1
    gravity: list = field(default_factory=lambda: [0., 0., -9.81])
    up_axis: int = 1
    use_gpu_pipeline: bool = True
    physx: PhysxParams = field(default_factory=PhysxParams)


def parse_sim_params(args, cfg):
    """helpers.py:79-101."""
    sp = SimParams()
    if args.physics_engine == SIM_PHYSX:
        sp.physx.use_gpu = args.use_gpu
        sp.physx.num_subscenes = args.subscenes
    sp.use_gpu_pipeline = args.use_gpu_pipeline
    if "sim" in cfg:
        for k, v in cfg["sim"].items():
            if k == "physx":
                for pk, pv in v.items():
                    setattr(sp.physx, pk, pv)
            else:
                setattr(sp, k, v)
    if args.physics_engine == SIM_PHYSX and args.num_threads > 0:
        sp.physx.num_threads = args.num_threads
    return sp


def get_load_path(root, load_run=-1, checkpoint=-1):
    """helpers.py:103-125: last run (lexicographic), highest-numbered model_*.pt."""
    try:
        runs = os.listdir(root)
        runs.sort()
        if 'exported' in runs:
            runs.remove('exported')
        last_run = os.path.join(root, runs[-1])
    except Exception:
        raise ValueError("No runs in this directory: " + root)
    load_run = last_run if load_run == -1 else os.path.join(root, load_run)
    if checkpoint == -1:
        models = [f for f in os.listdir(load_run) if 'model' in f]
        models.sort(key=lambda m: '{0:0>15}'.format(m))
        model = models[-1]
    else:
        model = "model_{}.pt".format(checkpoint)
    return os.path.join(load_run, model)


def update_cfg_from_args(env_cfg, cfg_train, args):
    """helpers.py:127-150."""
    if env_cfg is not None:
        if args.num_envs is not None:
            env_cfg.env.num_envs = args.num_envs
    if cfg_train is not None:
        if args.seed is not None:
            cfg_train.seed = args.seed
        if args.max_iterations is not None:
            cfg_train.runner.max_iterations = args.max_iterations
        if args.resume:
            cfg_train.runner.resume = args.resume
        if args.experiment_name is not None:
            cfg_train.runner.experiment_name = args.experiment_name
        if args.run_name is not None:
            cfg_train.runner.run_name = args.run_name
        if args.load_run is not None:
            cfg_train.runner.load_run = args.load_run
        if args.checkpoint is not None:
            cfg_train.runner.checkpoint = args.checkpoint
    return env_cfg, cfg_train


def _parse_device_str(s):
    parts = s.split(":")
    return parts[0].lower(), (int(parts[1]) if len(parts) > 1 else 0)


def get_args(argv=None):
    p = argparse.ArgumentParser(description="RL Policy")
    p.add_argument("--task", type=str, default="pointfoot_rough")
    p.add_argument("--resume", action="store_true", default=False)
    p.add_argument("--experiment_name", type=str)
    p.add_argument("--run_name", type=str)
    p.add_argument("--load_run", type=str)
    p.add_argument("--checkpoint", type=int)
    p.add_argument("--headless", action="store_true", default=False)
    p.add_argument("--horovod", action="store_true", default=False)
    p.add_argument("--rl_device", type=str, default="cuda:0")
    p.add_argument("--num_envs", type=int)
    p.add_argument("--seed", type=int)
    p.add_argument("--max_iterations", type=int)
    # flags contributed by gymutil.parse_arguments in the reference
    p.add_argument("--sim_device", type=str, default="cuda:0")
    p.add_argument("--pipeline", type=str, default="gpu")
    p.add_argument("--graphics_device_id", type=int, default=0)
    g = p.add_mutually_exclusive_group()
    g.add_argument("--flex", action="store_true")
    g.add_argument("--physx", action="store_true")
    p.add_argument("--num_threads", type=int, default=0)
    p.add_argument("--subscenes", type=int, default=0)
    p.add_argument("--slices", type=int)
    args = p.parse_args(argv)
    args.sim_device_type, args.compute_device_id = _parse_device_str(args.sim_device)
    pipeline = args.pipeline.lower()
    assert pipeline in ("cpu", "gpu", "cuda"), f"Invalid pipeline '{args.pipeline}'. Should be either cpu or gpu."
    args.use_gpu_pipeline = pipeline in ("gpu", "cuda")
    if args.sim_device_type != "cuda" and args.flex:
        args.sim_device, args.sim_device_type, args.compute_device_id = "cuda:0", "cuda", 0
    if args.sim_device_type != "cuda" and pipeline == "gpu":
        args.pipeline, args.use_gpu_pipeline = "CPU", False
    args.physics_engine = SIM_FLEX if args.flex else SIM_PHYSX
    args.use_gpu = args.sim_device_type == "cuda"
    if args.slices is None:
        args.slices = args.subscenes
    # name alignment (helpers.py:173-177)
    args.sim_device_id = args.compute_device_id
    args.sim_device = args.sim_device_type
    if args.sim_device == 'cuda':
        args.sim_device += f":{args.sim_device_id}"
    return args


def export_policy_as_jit(actor_critic, path):
    """helpers.py:180-190 (feed-forward branch; the recurrent exporter belongs to ActorCriticRecurrent, not built)."""
    if hasattr(actor_critic, 'memory_a'):
        raise NotImplementedError("recurrent policies are outside the pointfoot hot path")
    os.makedirs(path, exist_ok=True)
    path = os.path.join(path, 'policy_1.pt')
    model = copy.deepcopy(actor_critic.actor).to('cpu')
    torch.jit.script(model).save(path)
