"""bench.py's launcher logic, checked without a GPU: `--gpus N` must either run N ranks or fail -- never report a
1-GPU number for an N-GPU request (round-1 finding) --, and the CPU-baseline build tuned for the host computes the
same bits as the portable parity build."""
import os
import subprocess
import sys

import numpy as np

import kfx_testlib as T

BENCH = os.path.join(T.ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "KFX_BENCH_BACKEND")}
    env.update(kw)
    return env


def test_world_size_mismatch_is_an_error():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "8"], capture_output=True, text=True, timeout=120, env=_env(WORLD_SIZE="1"))
    assert out.returncode != 0 and "--gpus 8" in out.stderr and not out.stdout.strip()


def test_gpus_flag_spawns_rank_processes():
    """No GPU here, so the ranks stop at their first check -- what matters is that `--gpus 2` started two of them."""
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         timeout=600, env=_env())
    assert out.returncode != 0
    assert (out.stderr + out.stdout).count("bench.py needs a GPU") >= 2, out.stderr[-2000:]
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_native_baseline_build_computes_the_same_bits():
    import oracle
    from kangaroo_amd import scenes

    def run():
        N, w, h = 32, 80, 60
        bmin, bmax, near, far = scenes.SCENES["room"]
        K = scenes.intrinsics(w, h)
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        vol = oracle.Volume(N, N, N, bmin, bmax)
        oracle.sdf_reset(vol, float("nan"))
        f, vbo, nrm = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h, channels=4)
        rd, rn, ri = oracle.Image(w, h), oracle.Image(w, h, channels=4), oracle.Image(w, h)
        for i in range(2):
            T_wc = scenes.orbit_pose(i, 30)
            raw = oracle.Image.from_numpy(scenes.render_depth("room", w, h, T_wc, K))
            oracle.bilateral(f, raw, nthreads=2, **scenes.BILATERAL)
            oracle.depth_to_vbo(vbo, f, K)
            oracle.normals_from_vbo(nrm, vbo)
            oracle.sdf_fuse(vol, f, nrm, scenes.se3_inverse(T_wc), K, tr, scenes.MAX_W, scenes.MIN_COS_THETA, nthreads=2)
            oracle.raycast_sdf(rd, rn, ri, vol, T_wc, K, near, far, tr, True, nthreads=2)
        return [a.copy() for a in (f.data, nrm.data, vol.data, rd.data, rn.data, ri.data)]

    portable = run()
    so, lib_ = oracle._SO_OVERRIDE, oracle._LIB
    try:
        desc = oracle.use_native_build()
        native = run()
    finally:
        oracle._SO_OVERRIDE, oracle._LIB = so, lib_
    assert "march=native" in desc, desc
    for a, b in zip(portable, native):
        assert np.array_equal(a, b, equal_nan=True)
