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


def test_pmc_traffic_is_reported_only_for_the_kernels_it_was_taken_on(tmp_path, monkeypatch):
    """bench.py's `roofline.traffic` is a committed counter figure (PMC passes cannot run inside the timed region): the traffic
    file records the digest of the kernel sources it was taken on (kfx_kernel_source_id, scripts/make_pmc_traffic.py), and a
    figure of other kernels than the loaded library's comes back as None with the reason (round-4 verdict, item 8)."""
    import json
    sys.path.insert(0, T.ROOT)
    import bench
    from kangaroo_amd import _lib
    L = _lib.load()
    ids = {fam: L.kfx_kernel_source_id(fam.encode()).decode() for fam in ("fuse", "raycast")}
    assert all(len(v) == 16 for v in ids.values()) and L.kfx_kernel_source_id(b"nothing") == b""
    good = {"_kfx_version": int(L.kfx_version()), "_kernel_source_id": ids, "full_fast": {"traffic_bytes": 123}, "raycast_room_fast": {"traffic_bytes": 456}}
    stale = dict(good, _kernel_source_id=dict(ids, fuse="0123456789abcdef"))
    old = {"full_fast": {"traffic_bytes": 789}}   # a file from before the digests were recorded
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "PMC_TRAFFIC_FILES", ("a.json", "b.json"))
    (tmp_path / "a.json").write_text(json.dumps(good))
    t, src = bench.pmc_traffic("full_fast")
    assert t == 123 and "a.json" in src and ids["fuse"] in src
    assert bench.pmc_traffic("raycast_room_fast")[0] == 456
    (tmp_path / "a.json").write_text(json.dumps(stale))
    t, src = bench.pmc_traffic("full_fast")
    assert t is None and "not reported" in src and "0123456789abcdef" in src
    assert bench.pmc_traffic("raycast_room_fast")[0] == 456            # the other family's figure is still current
    (tmp_path / "b.json").write_text(json.dumps(good))                  # ... and a current file further down the list is used
    assert bench.pmc_traffic("full_fast")[0] == 123
    (tmp_path / "a.json").write_text(json.dumps(old))
    (tmp_path / "b.json").unlink()
    t, src = bench.pmc_traffic("full_fast")
    assert t is None and "unrecorded revision" in src
    assert bench.pmc_traffic("no_such_key") == (None, None)


def test_cpu_baseline_is_a_median_of_three_blocks():
    """Round-5 verdict, item 6: `cpu_baseline.value` is the median of three blocks of full frames with the spread beside it (one
    8-second sample moved 19 % between two rounds on unchanged code).  A small volume here: the shape of the record and the
    arithmetic, not the figure."""
    import argparse
    sys.path.insert(0, T.ROOT)
    import bench
    import oracle
    so, lib_ = oracle._SO_OVERRIDE, oracle._LIB
    try:
        out = bench.cpu_baseline(argparse.Namespace(res=32, width=80, height=60), "room", 30)
    finally:
        oracle._SO_OVERRIDE, oracle._LIB = so, lib_
    assert out["kind"] == "port" and out["unit"] == "frames/s" and out["cores"] >= 1
    assert len(out["blocks_fps"]) == 3 and min(out["blocks_fps"]) > 0
    assert abs(out["value"] - float(np.median(out["blocks_fps"]))) < 1e-3
    assert abs(out["spread"] - (max(out["blocks_fps"]) - min(out["blocks_fps"])) / out["value"]) < 1e-3
    assert "median of 3 blocks of 6 full frames" in out["sample"] and out["single_thread"]["cores"] == 1
