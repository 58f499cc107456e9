"""The RCCL code paths on the one GPU a test box has: world size 1.  A degenerate gather, but everything that never ran
before runs: librccl is loaded, a communicator is created (ncclCommInitRank), ncclAllGather is enqueued on the caller's
stream, the overlapped form orders its two streams with events, and torch.distributed's nccl backend takes the same
tensors ("the RCCL code path has never executed anywhere")."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_rccl_world_size_one_native_and_torch():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_comm_world1_child.py"), str(_free_port())], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["uid_nonzero"]
    assert out["comm_rank_world"] == [0, 1]
    for k, v in out.items():
        if k.startswith(("native_", "torch_")):
            assert v is True, (k, out)
    assert {"native_allgather_world1", "native_decode_allgather_chunks1", "native_decode_allgather_chunks3",
            "native_decode_allgather_chunks4", "torch_nccl_allgather_world1", "torch_nccl_chunked_world1"} <= set(out)


def test_comm_fails_cleanly_without_a_gpu():
    """No GPU here: the id can still be asked for (RCCL present) or the call reports ALACGPU_ERR_COMM; nothing crashes, and
    creating a communicator without a context is a bad argument."""
    import ctypes as C
    import alac.net_amd as pkg

    L = pkg.lib()
    assert L.alacgpu_comm_create(None, None, 0, 1, C.byref(C.c_void_p())) == -1
    assert L.alacgpu_comm_rank(None) == -1 and L.alacgpu_comm_world(None) == 0
    assert L.alacgpu_allgather_pcm(None, None, None, 0, None) == -1
