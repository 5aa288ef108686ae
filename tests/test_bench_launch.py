"""bench.py as the driver runs it: `python bench.py --gpus N` has to become N ranks by itself (the reference has
no multi-GPU launcher at all -- legged_gym/scripts/train.py:41-44, the dead --horovod flag at legged_gym/utils/helpers.py:162).
The PARENT of such a run must never initialise the GPU (a process that has may not start the ranks by exec, and HIP state
does not survive into children): it parses the flags and starts `python -m torch.distributed.run` as a child BEFORE torch is
imported."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

import common

BENCH = os.path.join(common.ROOT, "bench.py")


def _probe(code):
    """run `code` in a fresh interpreter with bench.py loaded as a module (not __main__)"""
    pre = textwrap.dedent(f"""
        import importlib.util, sys, json
        sys.argv = ["bench.py"]
        spec = importlib.util.spec_from_file_location("bench_mod", {BENCH!r})
    """)
    return subprocess.run([sys.executable, "-c", pre + textwrap.dedent(code)], capture_output=True, text=True, timeout=300)


def test_parent_builds_the_torchrun_command_before_torch_is_imported():
    # the parent path = everything bench.py executes up to its `sys.exit(self_launch(...))`: a stub subprocess.Popen records the
    # command; torch must not be in sys.modules at that point, and the JSON line of "rank 0" is relayed on stdout
    code = """
        import subprocess, os, runpy, io
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            os.environ.pop(k, None)
        seen = {}
        class FakeProc:
            def __init__(self, cmd, **kw):
                seen["cmd"] = cmd; seen["env"] = kw.get("env"); seen["torch_loaded"] = "torch" in sys.modules
                self.stdout = io.StringIO('[Gloo] banner\\n{"metric": "m", "value": 1.0, "n_gpus": 8}\\n')
            def wait(self):
                return 0
        subprocess.Popen = FakeProc
        sys.argv = ["bench.py", "--gpus", "8", "--steps", "3", "--warmup", "1", "--master_port", "29617"]
        try:
            runpy.run_path(%r, run_name="__main__")
        except SystemExit as e:
            seen["rc"] = e.code
        seen["torch_after"] = "torch" in sys.modules
        seen["env"] = {k: seen["env"].get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY",)}
        print("PROBE" + json.dumps(seen))
    """ % BENCH
    r = _probe(code)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    seen = json.loads([ln for ln in lines if ln.startswith("PROBE")][0][5:])
    assert seen["rc"] == 0 and not seen["torch_loaded"] and not seen["torch_after"]
    assert seen["cmd"][1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
                                "--master-port", "29617"]
    assert seen["cmd"][9] == BENCH and seen["cmd"][10:] == ["--gpus", "8", "--steps", "3", "--warmup", "1", "--master_port", "29617"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # stdout of the parent: the relayed JSON line and nothing else from the ranks (the banner went to stderr)
    assert [ln for ln in lines if not ln.startswith("PROBE")] == ['{"metric": "m", "value": 1.0, "n_gpus": 8}']
    assert "[Gloo] banner" in r.stderr


def test_self_launch_decision_and_failure_codes():
    code = """
        m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)     # imports torch: this is the rank-side view
        a = m.parse(["--gpus", "4"])
        out = {"n4": m.needs_self_launch(a, {}), "n4_torchrun": m.needs_self_launch(a, {"WORLD_SIZE": "4", "RANK": "0"}),
               "n1": m.needs_self_launch(m.parse([]), {}), "n1_forced": m.needs_self_launch(m.parse(["--self_launch"]), {}),
               "cmd": m.self_launch_command(m.parse(["--gpus", "1", "--self_launch"]), ["--gpus", "1", "--self_launch"], 1234)}
        print("PROBE" + json.dumps(out))
    """
    r = _probe(code)
    assert r.returncode == 0, r.stderr
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("PROBE")][0][5:])
    assert out["n4"] and not out["n4_torchrun"] and not out["n1"] and out["n1_forced"]
    assert "--self_launch" not in out["cmd"] and "--nproc-per-node=1" in out["cmd"]      # the child must not launch again


def test_parent_relays_a_failing_child_exit_code(tmp_path):
    # a child that dies (here: torchrun cannot even start the script's ranks because --gpus disagrees with nothing -- we use a bad
    # flag) must surface as a non-zero exit code of bench.py with nothing on stdout
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "no_such_backend",
                        "--no_cpu_baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "starting ranks" in r.stderr


@pytest.mark.gpu
def test_self_launched_single_rank_rccl_bench():
    """The self-launch path end to end on the one GPU of a test box: parent -> torchrun -> ONE rank on the nccl (= RCCL) backend
    bound to cuda:0 by device_id, every collective of the N>1 path issued (PF_DP_SINGLE_RANK_COLLECTIVES=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PF_DP_SINGLE_RANK_COLLECTIVES"] = "1"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "2", "--warmup", "1", "--self_launch", "--no_cpu_baseline", "--no_alt"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out_lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(out_lines) == 1, out_lines
    line = json.loads(out_lines[0])
    assert line["n_gpus"] == 1 and line["collectives"].startswith("nccl, ") and line["ranks_in_lockstep"] is True
    assert line["config"]["finite"] and line["value"] > 0
    assert "can cause a hang" not in r.stderr      # the warning RCCL prints for a process group without device_id
