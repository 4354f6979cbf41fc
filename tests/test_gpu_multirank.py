"""The sharded frame pipeline end to end on the GPU with TWO ranks: tile dealing, sharded ray generation, trace,
shade to float RGBA into the gather slab, asynchronous gather, scatter on rank 0 -- and the gathered frame must be
bit-identical to the same frame rendered by one rank.  The ranks share the box's single GPU, so the process group
is gloo (RCCL refuses two ranks on one device); the kernels and the frame-end code are the ones bench.py runs."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = os.path.join(ROOT, "scripts", "dist_bit_identity_worker.py")   # (also a step of scripts/first_node_run.py)


@pytest.mark.parametrize("world", [2, 8])
def test_two_ranks_gathered_frame_equals_single_rank_frame(world):
    """world = 8: the 160 x 96 frame has 15 tiles -- one rank is dealt NONE (an empty shard: no pixels, no rays, no arrays;
    found by the node script's stand-in run in round 5: the shade call refused its NULL object-id array)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, BHG_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.pop("BHG_DISTINCT", None)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), WORKER],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "MULTIRANK_OK" in out.stdout and f"'world': {world}" in out.stdout
