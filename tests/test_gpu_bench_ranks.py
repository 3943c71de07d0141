"""bench.py as the driver launches it for N > 1: one process per rank under torch.distributed.run, barrier + max-over-ranks timing, rank 0 printing one
JSON line whose value is the whole job's rate.  A 1-GPU box has no second device, so the two ranks share GPU 0 and rendezvous over gloo
(--x-backend / --x-device are test hooks of bench.py; on the 8-GPU node the same code runs with the RCCL backend, one rank per GPU).  The decode path
has no data-path collective: ranks are independent replicas ("scaling": "weak")."""
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


HOOKS = ["--x-backend", "gloo", "--x-device", "0"]   # two ranks share GPU 0 over gloo


def test_two_ranks_one_json_line():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "96", "--warmup", "16", "--cpu-seconds", "0", "--streams", "0", "--config", "small"] + HOOKS
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 96 and j["warmup"] == 16 and j["scaling"] == "weak" and j["higher_is_better"] is True
    assert j["config"]["replicas"] == 2
    # whole-job rate = units of all ranks / max-over-ranks time
    assert abs(j["value"] - 2 * 96 / (j["ms_per_step"] * 96 / 1e3)) <= 1e-2 * j["value"]
    assert "roofline" in j and j["vs_baseline"] is None


def test_two_ranks_tensor_parallel_line():
    """`bench.py --config qwen3-32b --gpus 2` dispatches TP = 2 (BASELINE config 4) through the C++ host's graph with the kernel-side exchange; the two
    ranks share GPU 0 here, their receive areas cross the process boundary as IPC handles.  Cut to 2 layers / 8192 vocabulary rows (test hook)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "16", "--config", "qwen3-32b", "--x-tp-layers", "2", "--x-tp-vocab", "8192"] + HOOKS
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 64 and j["scaling"] == "strong" and j["config"]["tp"] == 2 and j["config"]["exchange"] == "p2p"
    assert abs(j["value"] - 64 / (j["ms_per_step"] * 64 / 1e3)) <= 1e-2 * j["value"]          # ONE sequence: tokens/s of the whole job
    assert "roofline" in j and j["vs_baseline"] is None


def test_plain_gpus_2_launches_two_ranks():
    """`python bench.py --gpus 2` run bare (no launcher around it, as the driver runs `--gpus 1`) starts the two rank processes itself -- the parent never touches the GPU --
    and relays ONE line with n_gpus == 2.  The TP leg the launcher adds for the default config is switched off here (its own test is above): --x-no-tp-leg."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "64", "--warmup", "16", "--cpu-seconds", "0", "--streams", "0", "--config", "small", "--x-no-tp-leg"] + HOOKS
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line: %r" % lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 64 and j["warmup"] == 16 and j["scaling"] == "weak" and j["config"]["replicas"] == 2
