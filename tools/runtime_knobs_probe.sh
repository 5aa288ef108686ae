run() { # name, env assignments...
  name=$1; shift
  env "$@" timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_alt > gpurun_out/s3_knob_$name.json 2> gpurun_out/s3_knob_$name.err || { echo "$name FAILED"; tail -3 gpurun_out/s3_knob_$name.err; return 0; }
  python3 - $name <<'P'
import json,sys
n=sys.argv[1]
d=json.loads(open(f"gpurun_out/s3_knob_{n}.json").read().strip().splitlines()[-1]); p=d["phase_ms"]["median"]
print(f"{n:28s} {d['median_value']/1e6:6.3f} M  it {p['iteration']:7.3f}  rollout {p['rollout']:6.3f}  update {p['ppo_update']:7.3f}  step {p['env_step_launch']*1e3:5.1f} us  finite {d['config'].get('finite')}")
P
}
run base0 PFX=0
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run optflush0 AMD_OPT_FLUSH=0
run hwq1 GPU_MAX_HW_QUEUES=1
run kacopt1 DEBUG_HIP_KERNARG_COPY_OPT=1
run kacopt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run skipcopy1 ROC_SKIP_KERNEL_ARG_COPY=1
run fgs1 ROC_USE_FGS_KERNARG=1
run fgs0 ROC_USE_FGS_KERNARG=0
run hdpwa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run base1 PFX=0
