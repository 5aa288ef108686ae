"""Probe (not a product path): what would software-pipelining the rollout over two half batches buy?
One env of N robots stepped as today, against TWO envs of N/2 robots whose rollouts are issued alternately on two HIP streams
-- half A's env step runs beside half B's policy forward and the other way round.  Both arrangements do the same work
(N x T env-steps, N x T policy rows); the two-env arrangement is what a split `pf_step` / `ppo_act` over env ranges would produce.
   python tools/halfbatch_probe.py [N] [rollouts]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = "cuda:0"
torch.cuda.set_device(0)


def start(env):
    obs = env.get_observations()
    cobs = env.get_privileged_observations()
    return obs, cobs if cobs is not None else obs


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(reps):
        fn()
    t_enq = (time.perf_counter() - t0) / reps
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps, t_enq * 1e3


env, runner, _ = bench.make("pointfoot_rough", N, dev)
state = list(start(env))
def one():
    state[0], state[1], _ = runner.rollout(state[0], state[1])
    runner.alg.storage.clear()
ms1, enq1 = timed(one, R)
T = runner.num_steps_per_env
print(f"one env of {N}: rollout {ms1:.3f} ms ({ms1 / T * 1e3:.1f} us per policy step), host enqueue {enq1:.2f} ms")
del env, runner

halves = [bench.make("pointfoot_rough", N // 2, dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
st = [list(start(h[0])) for h in halves]

def two(streams_used):
    algs = [h[1].alg for h in halves]
    envs = [h[0] for h in halves]
    for _ in range(T):
        for k in range(2):
            with torch.cuda.stream(streams_used[k]):
                actions = algs[k].act(st[k][0], st[k][1])
                obs, priv, rew, dones, infos = envs[k].step(actions)
                st[k][0], st[k][1] = obs, priv if priv is not None else obs
                algs[k].process_env_step(rew, dones, infos)
    for k in range(2):
        with torch.cuda.stream(streams_used[k]):
            algs[k].compute_returns(st[k][1])
            algs[k].storage.clear()

def on_two():
    cur = torch.cuda.current_stream()
    for s in streams: s.wait_stream(cur)
    two(streams)
    for s in streams: cur.wait_stream(s)
def on_one():
    cur = torch.cuda.current_stream()
    two([cur, cur])

ms2, enq2 = timed(on_two, R)
print(f"two envs of {N // 2}, alternating on two streams: {ms2:.3f} ms per rollout of both ({ms2 / T * 1e3:.1f} us per policy step), host enqueue {enq2:.2f} ms")
ms3, enq3 = timed(on_one, R)
print(f"two envs of {N // 2}, one stream (no overlap): {ms3:.3f} ms, host enqueue {enq3:.2f} ms")
print(f"two streams / one env: {ms2 / ms1:.3f}")
