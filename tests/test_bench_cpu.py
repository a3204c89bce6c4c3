"""CPU: the parts of bench.py's N > 1 path that need no GPU -- the launch line `python bench.py --gpus N` re-executes itself with
(torch.distributed.run, 127.0.0.1, one rank per GPU: the driver's own line), and the mismatches it refuses instead of silently running
a different job: more ranks than GPUs, WORLD_SIZE != --gpus, LOCAL_RANK beyond the visible GPUs, --share-gpu without gloo.  The ranks
themselves run under `-m gpu` (tests/test_hip_round5.py: two gloo ranks sharing the one GPU of the box)."""
import argparse
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ns(**kw):
    return argparse.Namespace(**dict(dict(gpus=1, share_gpu=False, dist_backend="nccl"), **kw))


def test_the_spawn_line_is_the_drivers_line():
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    cmd, env = bench.torchrun_command(_ns(gpus=8), argv, n_visible=8)
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[:i] == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                       "--master-port", cmd[i - 1]]
    assert 1024 < int(cmd[i - 1]) < 65536 and cmd[i + 1:] == argv            # a free port; the caller's own arguments, unchanged
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this host
    # the dry run: two ranks on one GPU are allowed only as such
    cmd, _ = bench.torchrun_command(_ns(gpus=2, share_gpu=True, dist_backend="gloo"), ["--gpus", "2", "--share-gpu", "--dist-backend", "gloo"], n_visible=1)
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    with pytest.raises(SystemExit, match="only 1 GPUs are visible"):
        bench.torchrun_command(_ns(gpus=2), [], n_visible=1)
    with pytest.raises(SystemExit, match="needs one visible GPU"):
        bench.torchrun_command(_ns(gpus=2, share_gpu=True, dist_backend="gloo"), [], n_visible=0)


def test_a_rank_checks_its_environment_against_the_arguments(monkeypatch):
    import bench
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    assert bench.check_rank_environment(_ns(), 1) == (1, 0, 0, 0)
    monkeypatch.setenv("WORLD_SIZE", "8"); monkeypatch.setenv("RANK", "5"); monkeypatch.setenv("LOCAL_RANK", "5")
    assert bench.check_rank_environment(_ns(gpus=8), 8) == (8, 5, 5, 5)                          # one rank per GPU: cuda:LOCAL_RANK
    assert bench.check_rank_environment(_ns(gpus=8, share_gpu=True, dist_backend="gloo"), 1) == (8, 5, 5, 0)   # dry run: every rank on cuda:0
    with pytest.raises(SystemExit, match="--gpus 4 but WORLD_SIZE=8"):
        bench.check_rank_environment(_ns(gpus=4), 8)
    with pytest.raises(SystemExit, match="LOCAL_RANK 5 but only 4 GPUs visible"):
        bench.check_rank_environment(_ns(gpus=8), 4)
    with pytest.raises(SystemExit, match="add --dist-backend gloo"):
        bench.check_rank_environment(_ns(gpus=8, share_gpu=True), 8)


def test_bench_exits_before_touching_a_device_when_the_job_does_not_fit():
    """The script itself, here where no GPU is visible: each mismatch is a non-zero exit with its message, never a traceback from CUDA."""
    def run(env, *args):
        e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=dict(e, **env), capture_output=True, text=True, timeout=300)
        return out.returncode, out.stdout + out.stderr
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    rc, msg = run({}, "--gpus", "2")
    assert rc != 0 and "--gpus 2 but only 0 GPUs are visible" in msg and "Traceback" not in msg
    rc, msg = run({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert rc != 0 and "--gpus 2 but WORLD_SIZE=3" in msg and "Traceback" not in msg
    rc, msg = run({"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, "--gpus", "2")
    assert rc != 0 and "LOCAL_RANK 1 but only 0 GPUs visible" in msg and "Traceback" not in msg
