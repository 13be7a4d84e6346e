"""The RCCL path itself, executed (VERDICT r02 missing #3): backend "nccl" (= RCCL on ROCm), world size 1 on the one-GPU
box, in a child process -- gather.SlotRing + gather.CudaRuntime + unina_infer_async for 64 frames on two streams, every
gathered slot byte-equal to a serial unina_infer of its frame. (The gloo tests cover the same loop's control flow at world
size 2 on CPU; what only hardware can show is that the collectives are accepted on the comm stream behind frame events.)"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slot_ring_over_rccl_world_size_1():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), "64"], capture_output=True, text=True,
                       env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_OK frames=64 gathers=16 world=1" in r.stdout
