"""Two processes, one rank each, both on the one GPU of the test box: the receive areas are exported / opened as IPC handles and the TP step runs
as it does across xGMI -- kernels of one process storing into memory another process polls.  The ids must equal the single-process NativeTP's
(and therefore the oracle's TP emulation, tests/test_gpu_tp.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import oracle_model
from koifish_amd import lib as L
from koifish_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_processes_exchange_through_ipc_mapped_areas(tmp_path):
    out = str(tmp_path / "ids.json")
    for r in range(2):   # a stale result of an earlier (timed-out) run must not pass for this one's
        if os.path.exists("%s.%d" % (out, r)):
            os.remove("%s.%d" % (out, r))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "_tp_ipc_child.py"), out]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = [json.load(open("%s.%d" % (out, r))) for r in range(2)]
    assert res[0] == res[1] and res[0]["eager"] == res[0]["graph"]
    cfg = dict(synth.CONFIGS["small"])
    raw = synth.raw_weights_numpy(cfg, 31, w_std=0.1)
    prompt = np.random.default_rng(2).integers(0, cfg["vocab"], size=10).astype(np.int32)
    om = oracle_model(cfg, raw, L.Q4, L.BF16, tp=2)
    assert res[0]["graph"] == om.generate(prompt.tolist(), 20)
