"""GPU: the N > 1 code path on RCCL, rehearsed on the one GPU of the test box (world_size 1: RCCL refuses two ranks on one
device).  This module sorts FIRST among the GPU tests on purpose: the pytest process has not touched the GPU when it
starts the ranks as child processes (torch.distributed.run), and it never execs itself.

* scripts/rccl_rank.py: BatchedLocalExpertOI.run() in a "nccl" process group through sharding.gather_arrays
  (gather="always") = the un-grouped run, table for table; a raw sharding.gather_results of device tensors; an all_reduce;
* bench.py --global-tiles 8192 (BASELINE configs[3]'s path: LPT shard, gather(v) to rank 0) launched the way the driver
  launches N > 1, and `python bench.py --gpus 1` launched plainly."""
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
    p = s.getsockname()[1]
    s.close()
    return p


def _gpu_touched():
    try:
        return any("kfd" in os.readlink(f"/proc/self/fd/{fd}") for fd in os.listdir("/proc/self/fd"))
    except OSError:
        return False


def _torchrun(args, timeout):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    return subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_orchestrator_and_gather_on_rccl(tmp_path):
    assert not _gpu_touched(), "this module must run before any test that opens the GPU in the pytest process"
    out = tmp_path / "verdict.json"
    p = _torchrun([os.path.join(ROOT, "scripts", "rccl_rank.py"), str(out)], 600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    v = json.load(open(out))
    assert v["backend"] == "nccl" and v["world"] == 1
    assert v["tables_equal"] and v["experts"] == 144 and v["preds"] == 144 * 6
    assert v["raw_gather_ok"] and v["all_reduce_max"] == 1.0


def test_bench_global_tiles_through_torchrun():
    assert not _gpu_touched()
    p = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--global-tiles", "8192", "--steps", "1", "--warmup", "1",
                   "--cpu-tiles", "0", "--no-host-leg", "--no-quality"], 900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["config"]["global_tiles"] == 8192
    assert r["value"] > 1000 and r["config"]["failed_tiles"] == 0 and "gather to rank 0" in r["config"]["parallelism"]
