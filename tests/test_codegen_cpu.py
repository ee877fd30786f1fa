"""Codegen guard (no GPU: hipcc cross-compiles gfx950): the voxel loops of the SdfFuse kernels must not drain vector
memory at the head of an iteration -- a register-allocation accident that cost the bit-exact kernel 11 % in round 2 with
an otherwise identical instruction sequence (scripts/check_fuse_codegen.py)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fuse_voxel_loops_do_not_wait_for_vector_memory():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_fuse_codegen.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 vector-memory waits" in out.stdout and ", 0 over the VGPR" in out.stdout
    # the tracked fast kernels' hand-written loads of the cached planes: nothing touches a destination register before the wait
    assert ", 0 instructions touching their registers before the wait" in out.stdout and " 0 hand-written" not in out.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_raycast_split_loads_are_not_touched_before_their_wait():
    """Round-3 advice: the class-table march requests a sample's cells in one hand-written block (RayF32::issue) and waits for
    them in a later one (finish); the compiler does not know the loads are in flight.  Nothing it emits in between may name
    the destination registers or spill through scratch / AGPRs (scripts/check_raycast_codegen.py)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_raycast_codegen.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert ", 0 hazards between request and wait" in out.stdout and " 0 split cell requests" not in out.stdout
    assert ", 0 with scratch" in out.stdout
