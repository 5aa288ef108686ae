"""Probe for VERDICT r2 item 7: what would splitting the rollout into two half-batches on two streams buy?  No library change:
two independent env + PPO pairs of N/2 envs each stand in for the two halves; their rollouts are issued interleaved from one
host thread on two HIP streams (stream B starts one policy-forward late, so that A's env step runs beside B's MLP), and
compared with (a) one pair of N envs on one stream, (b) the two half pairs back to back on ONE stream.
   python tools/overlap_probe.py [N] [iters]"""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointfoot_amd.envs import task_registry
from pointfoot_amd.utils import get_args

def make(n):
    args = get_args(["--task", "pointfoot_rough", "--headless", "--num_envs", str(n)])
    with contextlib.redirect_stdout(sys.stderr):
        env, _ = task_registry.make_env("pointfoot_rough", args=args)
        runner, _ = task_registry.make_alg_runner(env, "pointfoot_rough", args=args, log_root=None)
    env.episode_length_buf = torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length))
    return env, runner

def rollout_steps(envs, runners, obs, streams, T):
    """T policy steps of every (env, runner) pair, pair i on streams[i]; pairs are issued step by step, interleaved"""
    for t in range(T):
        for i, (env, r) in enumerate(zip(envs, runners)):
            with torch.cuda.stream(streams[i]):
                o, c = obs[i]
                a = r.alg.act(o, c)
                o, p, rew, d, info = env.step(a)
                r.alg.process_env_step(rew, d, info)
                obs[i] = (o, p)

def timed(fn, iters):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    T = 24
    full_env, full_r = make(N)
    halves = [make(N // 2) for _ in range(2)]
    cur = torch.cuda.current_stream()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    def reset_storage(rs):
        for r in rs: r.alg.storage.clear()
    fo = [(full_env.get_observations(), full_env.get_privileged_observations())]
    def full():
        rollout_steps([full_env], [full_r], fo, [cur], T); reset_storage([full_r])
    ho = [(e.get_observations(), e.get_privileged_observations()) for e, _ in halves]
    envs, rs = [e for e, _ in halves], [r for _, r in halves]
    def serial():
        rollout_steps(envs, rs, ho, [cur, cur], T); reset_storage(rs)
    def overlapped():
        sa.wait_stream(cur); sb.wait_stream(cur)
        # stagger: B's first policy forward waits for A's
        with torch.cuda.stream(sa):
            o, c = ho[0]; a = rs[0].alg.act(o, c)
            ev = torch.cuda.Event(); ev.record(sa)
            o, p, rew, d, info = envs[0].step(a); rs[0].alg.process_env_step(rew, d, info); ho[0] = (o, p)
        sb.wait_event(ev)
        with torch.cuda.stream(sb):
            o, c = ho[1]; a = rs[1].alg.act(o, c)
            o, p, rew, d, info = envs[1].step(a); rs[1].alg.process_env_step(rew, d, info); ho[1] = (o, p)
        rollout_steps(envs, rs, ho, [sa, sb], T - 1)
        cur.wait_stream(sa); cur.wait_stream(sb)
        reset_storage(rs)
    print(f"N = {N}, {T} policy steps per rollout (no GAE), ms per rollout:")
    print(f"  one batch of {N} envs, one stream          {timed(full, iters):7.3f}")
    print(f"  two halves of {N // 2}, same stream (serial)   {timed(serial, iters):7.3f}")
    print(f"  two halves of {N // 2}, two streams (staggered) {timed(overlapped, iters):7.3f}")
    # host-side cost of issuing one rollout (no GPU wait): is the loop launch-bound?
    torch.cuda.synchronize(); t0 = time.perf_counter(); full(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"  host time to ISSUE the one-batch rollout   {(t1 - t0) * 1e3:7.3f}")
    torch.cuda.synchronize(); t0 = time.perf_counter(); overlapped(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"  host time to ISSUE the two-stream rollout  {(t1 - t0) * 1e3:7.3f}")

if __name__ == '__main__':
    main()
