"""Multi-rank runs of the Z-slab pipeline on the GPU box: real HIP operators and composite / hand-over kernels under
real collectives.  The box has one GPU, so the ranks share it and the transport is gloo (RCCL refuses two ranks on one
device); everything else is the code path `bench.py --gpus N` takes.  See tests/mp_slab_gpu.py."""
import os
import socket
import subprocess
import sys

import pytest

import kfx_testlib as T

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,halo,raycast,mode", [(2, "recompute", "composite", ""), (2, "exchange", "exact", ""), (3, "exchange", "composite", ""),
                                                     (3, "recompute", "exact", ""), (2, "recompute", "exact", "tracking"),
                                                     (2, "exchange", "composite", "tracking")])
def test_gpu_slab_pipeline_ranks_sharing_one_gpu(world, halo, raycast, mode):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(T.ROOT, "tests", "mp_slab_gpu.py"), halo, raycast] + ([mode] if mode else [])
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=T.ROOT)
    assert out.returncode == 0 and out.stdout.count("MP_OK") == world, out.stdout[-3000:] + out.stderr[-3000:]
