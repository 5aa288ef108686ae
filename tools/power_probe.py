"""Is the PPO update power-limited?  Samples `rocm-smi --showpower --showclocks` five times a second while the GPU runs, first, 3 s of PPO
updates back to back (GEMM-bound), then 3 s of rollouts (step kernel + small launches: latency-bound).  profiles/r3_power_probe.txt.
   python tools/power_probe.py"""
import subprocess, threading, time, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
samples = []
stop = False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.time(), out.strip()[:600]))
        except Exception as ex:
            samples.append((time.time(), repr(ex)))
        time.sleep(0.2)
env, runner, cfg = B.make("pointfoot_rough", 4096, "cuda:0")
obs = env.get_observations(); cobs = env.get_privileged_observations()
obs, cobs, _ = runner.rollout(obs, cobs); runner.alg.update(); torch.cuda.synchronize()
t = threading.Thread(target=poll); t.start()
time.sleep(1.0)
phase = []
t0 = time.time()
# 3 s of updates only (GEMM-bound), then 3 s of rollouts only (step-kernel / latency-bound)
while time.time() - t0 < 3.0:
    obs, cobs, _ = runner.rollout(obs, cobs)
    for _ in range(10): 
        runner.alg.storage.step = runner.num_steps_per_env
        runner.alg.update()
    torch.cuda.synchronize()
t1 = time.time()
while time.time() - t1 < 3.0:
    obs, cobs, _ = runner.rollout(obs, cobs); runner.alg.storage.clear() if hasattr(runner.alg.storage, "clear") else None
    torch.cuda.synchronize()
t2 = time.time()
time.sleep(0.5)
stop = True; t.join()
for ts, s in samples:
    tag = "idle" if ts < t0 else ("update" if ts < t1 else ("rollout" if ts < t2 else "idle"))
    print(tag, round(ts - t0, 2), s[:400].replace("\n", " "))
