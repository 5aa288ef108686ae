#!/bin/bash
# A/B of the policy step with its heads inside the env-step launch (PF_FUSED_POLICY_STEP=1) against the two-launch form, one box, alternating.
# usage (GPU box): bash tools/fused_step_ab.sh [task] [num_envs]
T=${1:-pointfoot_rough}; N=${2:-4096}
for v in 0 1 0 1 0 1; do
  PF_FUSED_POLICY_STEP=$v python3 bench.py --task $T --num_envs $N --steps 20 --warmup 5 --no_cpu_baseline --no_alt 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['phase_ms']['median']
print('PF_FUSED_POLICY_STEP=$v  %.3f M  iteration %.3f  rollout %.3f  update %.3f  step launch %.1f us  finite %s' % (d['median_value']/1e6, p['iteration'], p['rollout'], p['ppo_update'], p['env_step_launch']*1e3, d['config'].get('finite')))"
done
