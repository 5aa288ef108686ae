"""Does a smaller row count per backward pass pay (activations of one pass resident in the 256 MiB Infinity Cache)?

Times PPO.update of the headline configuration (pointfoot_rough, 4096 envs x 24 steps = 98 304 rows, 5 epochs) with the rows of an
epoch cut into 4 (the configuration: 24 576 rows, ~400 MB of activations + gradients per pass), 8, 16 and 32 minibatches.  The arithmetic
per epoch is the same; what changes is the working set of one forward / backward pass, the tile count per launch and the number of launches.
If the time per EPOCH falls with the row count, processing a 24 576-row minibatch as row chunks (same bits: the split-K slabs of the weight
gradients already cut the rows) would pay; if it is flat or rises, it would not.

usage (GPU box): python3 tools/minibatch_rows_probe.py [reps]
"""
import contextlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from pointfoot_amd.envs import task_registry
    from pointfoot_amd.utils import get_args
    dev = "cuda:0"
    args = get_args(["--task", "pointfoot_rough", "--headless", "--num_envs", "4096", "--sim_device", dev, "--rl_device", dev])
    with contextlib.redirect_stdout(sys.stderr):
        env, _ = task_registry.make_env(name="pointfoot_rough", args=args)
    rows = {}
    for nmb in (4, 8, 16, 32, 4):
        with contextlib.redirect_stdout(sys.stderr):
            _, train_cfg = task_registry.get_cfgs("pointfoot_rough")
            train_cfg.algorithm.num_mini_batches = nmb
            runner, _ = task_registry.make_alg_runner(env=env, name="pointfoot_rough", args=args, train_cfg=train_cfg, log_root=None)
        obs, cobs = env.get_observations(), env.get_privileged_observations()
        ms = []
        for i in range(reps + 3):
            obs, cobs, _ = runner.rollout(obs, cobs)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            runner.alg.update()
            e.record()
            torch.cuda.synchronize()
            if i >= 3:
                ms.append(s.elapsed_time(e))
        rows.setdefault(nmb, []).append(float(np.median(ms)))
        print(f"num_mini_batches {nmb:3d}  rows per pass {98304 // nmb:6d}   update median {np.median(ms):7.3f} ms   "
              f"per pass {1e3 * np.median(ms) / (5 * nmb):7.1f} us   per 24576 rows {1e3 * np.median(ms) / 20:7.1f} us", flush=True)
        del runner
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
