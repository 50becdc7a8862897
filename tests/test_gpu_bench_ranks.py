"""bench.py at world size 2, the way the scaling run launches it (torch.distributed.run, one process per rank).

A box with one GPU rehearses it: both ranks on device 0 (RTMI_ONE_GPU) and gloo as the collective backend
(RCCL cannot connect two ranks on one device); with two or more GPUs the same test also runs over RCCL on
distinct devices. Checked: the JSON line carries one {kernel_ms, reduce_ms} record per rank and the reduced
frame adds up to what the ranks rendered (bench.py asserts it and reports it as reduce_check)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run(split, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "96", "--height", "64", "--spp", "24", "--split", split, "--no-cpu-baseline", "--no-other-configs"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("split", ["tile", "sample"])
def test_two_ranks_on_one_gpu_over_gloo(split):
    out = _run(split, {"RTMI_ONE_GPU": "1", "RTMI_DIST_BACKEND": "gloo"})
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert [r["rank"] for r in out["per_rank"]] == [0, 1]
    assert all(r["kernel_ms"] > 0 and r["reduce_ms"] >= 0 for r in out["per_rank"])
    assert out["reduce_check"]["ok"] is True


def test_two_ranks_on_two_gpus_over_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: RCCL between two ranks needs two devices")
    out = _run("tile", {})
    assert out["n_gpus"] == 2 and out["reduce_check"]["ok"] is True
    assert "RCCL" in out["config"]["parallelism"]
