"""CPU-side checks of the drop-in boundary: libkfx.so loads without a GPU, exports every
symbol include/kfx.h declares, mirrors the reference container layouts, and rejects bad
arguments before touching the device (no compute calls here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np

import kfx_testlib as T
from kangaroo_amd import _lib

HEADER = os.path.join(T.ROOT, "include", "kfx.h")


def declared_symbols(headers=("kfx.h", "kfx_extras.h")):
    """kfx.h = the path; kfx_extras.h = the operators of the same five reference headers that SURVEY 2 marks out of scope (same library)."""
    names = set()
    for h in headers:
        src = open(os.path.join(T.ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(kfx_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 18
    path_only, extras = declared_symbols(("kfx.h",)), declared_symbols(("kfx_extras.h",))
    assert len(path_only) <= 80 and 8 <= len(extras) <= 16 and not set(path_only) & set(extras)   # (round-5 verdict, item 7: kfx.h is the path)
    for n in names:
        assert hasattr(L, n), "libkfx.so does not export %s" % n
        assert n in _lib.SIGNATURES, "python binding missing for %s" % n
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    exported = set(re.findall(r" T (kfx_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_debug_and_slab_headers_are_exported_too():
    """include/kfx_slab.h (multi-GPU slab partition) declares symbols of libkfx.so, except the RCCL transport, which lives in
    libkfx_rccl.so so that libkfx.so does not depend on librccl; include/kfx_debug.h (measurement aids, arithmetic
    self-checks) declares symbols of libkfx_debug.so only: the product library exports none of them."""
    def decl(path):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(T.ROOT, "include", path)).read(), flags=re.S)
        return set(re.findall(r"\b(kfx_[a-z0-9_]+)\s*\(", src))

    def exported(lib):
        out = subprocess.check_output(["nm", "-D", "--defined-only", lib]).decode()
        return set(re.findall(r" T (kfx_[a-z0-9_]+)", out))
    main = exported(_lib.LIB_PATH)
    rccl_lib = os.path.join(os.path.dirname(_lib.LIB_PATH), "libkfx_rccl.so")
    assert os.path.exists(rccl_lib), "libkfx_rccl.so not built"
    rccl = exported(rccl_lib)
    dbg, slab = decl("kfx_debug.h"), decl("kfx_slab.h")
    dbg_lib = os.path.join(os.path.dirname(_lib.LIB_PATH), "libkfx_debug.so")
    assert os.path.exists(dbg_lib), "libkfx_debug.so not built"
    assert len(dbg) >= 4 and dbg <= exported(dbg_lib)
    assert not any(n.startswith("kfx_debug") for n in main), "measurement aids inside the product library"
    assert _lib.load_debug() is not None   # resolves its libkfx.so dependencies without a GPU
    assert "kfx_comm_create_rccl" in slab and len(slab) >= 7
    assert slab - {"kfx_comm_create_rccl"} <= main and "kfx_comm_create_rccl" in rccl
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "rccl" not in needed
    assert "librccl" in subprocess.check_output(["readelf", "-d", rccl_lib]).decode()


def test_slab_layout_and_argument_checks():
    """kfx_slab_layout_init: contiguous owned ranges that tile the volume, ghost planes clipped at the ends, the local box
    from VoxelPositionInUnits of the whole volume; bad arguments are rejected (host-only code, no GPU needed)."""
    L = _lib.load()

    class Layout(C.Structure):
        _fields_ = [("full_d", C.c_size_t), ("full_zmin", C.c_float), ("full_zmax", C.c_float), ("rank", C.c_int), ("world", C.c_int),
                    ("ghost", C.c_int), ("z0", C.c_size_t), ("z1", C.c_size_t), ("s0", C.c_size_t), ("s1", C.c_size_t),
                    ("local_zmin", C.c_float), ("local_zmax", C.c_float)]
    L.kfx_slab_layout_init.argtypes = [C.POINTER(Layout), C.c_size_t, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
    from kangaroo_amd.pipeline import slab_range
    f = np.float32
    for d, world, ghost in ((512, 8, 2), (100, 3, 2), (64, 5, 1), (17, 4, 3)):
        prev = 0
        for r in range(world):
            lay = Layout()
            assert L.kfx_slab_layout_init(C.byref(lay), d, 2.0, 4.0, r, world, ghost) == 0
            assert (lay.z0, lay.z1) == slab_range(d, r, world) and lay.z0 == prev
            prev = lay.z1
            assert lay.s0 == max(lay.z0 - ghost, 0) and lay.s1 == min(lay.z1 + ghost, d)
            assert f(lay.local_zmin) == f(2.0) + (f(4.0) - f(2.0)) * f(lay.s0) / f(d - 1)
            assert f(lay.local_zmax) == f(2.0) + (f(4.0) - f(2.0)) * f(lay.s1 - 1) / f(d - 1)
        assert prev == d
    lay = Layout()
    assert L.kfx_slab_layout_init(C.byref(lay), 16, 0.0, 1.0, 0, 8, 3) == -4      # slabs thinner than the ghost width
    assert L.kfx_slab_layout_init(C.byref(lay), 16, 0.0, 1.0, 8, 8, 1) == -4 and L.kfx_slab_layout_init(None, 16, 0.0, 1.0, 0, 1, 1) == -1
    assert L.kfx_slab_exchange_halos(None, None, None, None) == -1 and L.kfx_slab_composite(None, None, None, None, None, None, None) == -1


def test_library_contains_gfx950_code_object():
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k_sdf_fuse" in blob and b"k_raycast_sdf" in blob


def test_struct_layouts_match_reference_containers():
    z = np.load(os.path.join(T.GOLDEN, "ref_helper_vectors.npz"))
    sizeof = z["sizeof"].tolist()  # measured on the reference headers by make_golden.py
    assert C.sizeof(_lib.KfxImage) == sizeof[0] == 32
    assert C.sizeof(_lib.KfxVolume) == sizeof[2] == 72
    assert _lib.KfxImage.pitch.offset == 0 and _lib.KfxImage.ptr.offset == 8
    assert _lib.KfxImage.w.offset == 16 and _lib.KfxImage.h.offset == 24
    assert _lib.KfxVolume.img_pitch.offset == 32 and _lib.KfxVolume.d.offset == 40
    assert _lib.KfxVolume.boxmin.offset == 48 and _lib.KfxVolume.boxmax.offset == 60


def test_argument_errors_are_reported_not_fatal():
    L = _lib.load()
    assert L.kfx_version() == 1
    K = (C.c_float * 4)(500, 500, 31.5, 23.5)
    Tm = (C.c_float * 12)(1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0)
    vol = _lib.KfxVolume(0, None, 8, 8, 0, 8)
    img = _lib.KfxImage(0, None, 8, 8)
    assert L.kfx_sdf_fuse(C.byref(vol), C.byref(img), C.byref(img), Tm, K, 0.1, 100.0, 0.1, 0, None) == -1
    assert b"KFX_E_NULL" in L.kfx_last_error_string()
    assert L.kfx_raycast_sdf(None, None, None, None, Tm, K, 0.1, 1.0, 0.1, 1, None) == -1
    assert L.kfx_normals_from_vbo(C.byref(img), C.byref(img), None) == -1
    assert L.kfx_bilateral_f32(C.byref(img), C.byref(img), 1.5, 0.1, 3, 0.2, 1, None) == -1
    # misaligned / undersized pitch on a fake non-null pointer: rejected before any launch
    bad = _lib.KfxVolume(60, 0x1000, 8, 8, 480, 8)
    assert L.kfx_sdf_reset(C.byref(bad), 0.0, None) == -2
    bad2 = _lib.KfxVolume(68, 0x1000, 8, 8, 68 * 8, 8)
    assert L.kfx_sdf_reset(C.byref(bad2), 0.0, None) == -3
    assert L.kfx_error_name(-3) == b"KFX_E_ALIGN" and L.kfx_error_name(0) == b"ok"
    p, pitch = C.c_void_p(), C.c_size_t()
    assert L.kfx_alloc_pitched(C.byref(p), C.byref(pitch), 0, 4) == -2
    assert L.kfx_free(None) == 0


def test_every_compute_entry_point_rejects_null_arguments():
    """Every entry point that takes container pointers validates them before any HIP call: all-NULL arguments
    return a negative KFX_E_* code (never a crash, never a launch) -- checked here without a GPU."""
    L = _lib.load()
    skip = {"kfx_version", "kfx_device_count", "kfx_last_error_string", "kfx_error_name", "kfx_free", "kfx_free_host",
            "kfx_set_math_mode", "kfx_get_math_mode", "kfx_stream_synchronize", "kfx_alloc_host", "kfx_memcpy_2d", "kfx_set_device", "kfx_sdf_summary_destroy", "kfx_frame_destroy"}
    checked = 0
    for name, (restype, argtypes) in sorted(_lib.SIGNATURES.items()):
        if name in skip or restype is not C.c_int:
            continue
        args = []
        for a in argtypes:
            if a in (C.c_float, C.c_double):
                args.append(a(0.5))
            elif a in (C.c_int, C.c_uint, C.c_size_t, C.c_longlong, C.c_ushort, C.c_ulonglong):
                args.append(a(0))
            else:
                args.append(None)       # pointers: NULL
        rc = getattr(L, name)(*args)
        assert rc < 0, "%s accepted NULL arguments (returned %d)" % (name, rc)
        assert L.kfx_last_error_string()
        checked += 1
    assert checked >= 35


def test_product_never_imports_the_oracle():
    """The product path must not route through oracle/ (tier rule 3)."""
    pkg = os.path.join(T.ROOT, "kangaroo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', txt), f
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "libkfx_oracle" not in txt and "libkfx_refhdr" not in txt, f
    for f in os.listdir(os.path.join(T.ROOT, "include")):
        p = os.path.join(T.ROOT, "include", f)
        if os.path.isfile(p):
            assert not re.search(r'#\s*include\s*[<"][^>"]*oracle', open(p).read())


def _rccl_lib():
    lib = os.path.join(os.path.dirname(_lib.LIB_PATH), "libkfx_rccl.so")
    try:
        R = C.CDLL(lib)
    except OSError as e:   # librccl not loadable on this host
        import pytest
        pytest.skip("libkfx_rccl.so not loadable here: %r" % (e,))
    R.kfx_comm_create_rccl.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int]
    R.kfx_rccl_launch_nonce.restype = C.c_ulonglong
    R.kfx_rccl_process_start.restype = C.c_longlong
    return lib, R


_RDV_ENV = ("TORCHELASTIC_RUN_ID", "MASTER_PORT", "SLURM_JOB_ID", "SLURM_STEP_ID", "KFX_RDV_SLACK_S", "KFX_RDV_PARENT")


def _rdv_payload(nonce):
    import struct
    return b"KFXRDV1\0" + struct.pack("<Q", nonce) + bytes(128)


def test_rccl_rendezvous_rejects_stale_foreign_and_linked_files(tmp_path, monkeypatch):
    """kfx_comm_create_rccl, ranks > 0 (comm_rccl.cpp): a rendezvous file left by an earlier run (older than this process), one
    written for another launch (different nonce) and a symlink are all ignored -- the call times out with KFX_E_RANGE instead
    of handing a foreign ncclUniqueId to ncclCommInitRank (which would hang).  Host-only: no GPU, no RCCL call is reached."""
    import time
    lib, R = _rccl_lib()
    comm = C.create_string_buffer(256)
    monkeypatch.setenv("KFX_RUN_ID", "this-launch")
    for k in _RDV_ENV:
        monkeypatch.delenv(k, raising=False)
    mine = R.kfx_rccl_launch_nonce()   # the environment's values + this process's parent (the launcher of a real run)
    monkeypatch.setenv("KFX_RUN_ID", "another-launch")
    other = R.kfx_rccl_launch_nonce()
    monkeypatch.setenv("KFX_RUN_ID", "this-launch")
    assert mine != other and mine & 1 and R.kfx_rccl_launch_nonce() == mine

    path = tmp_path / "id"
    # (0) control: fresh, right nonce -> accepted
    path.write_bytes(_rdv_payload(mine))
    assert R.kfx_rccl_rendezvous_probe(str(path).encode()) == 1
    # (1) right nonce, but older than this process
    os.utime(path, (time.time() - 3600, time.time() - 3600))
    assert R.kfx_comm_create_rccl(comm, 1, 2, str(path).encode(), 1) == -4   # KFX_E_RANGE: timed out
    # (2) fresh, but another launch's nonce
    path.write_bytes(_rdv_payload(other))
    assert R.kfx_comm_create_rccl(comm, 1, 2, str(path).encode(), 1) == -4
    # (3) a symlink to a fresh, well-formed file
    real = tmp_path / "real"
    real.write_bytes(_rdv_payload(mine))
    path.unlink()
    path.symlink_to(real)
    assert R.kfx_comm_create_rccl(comm, 1, 2, str(path).encode(), 1) == -4


def _probe_child(lib, path, env, sleep_s=0.0):
    """A fresh process (a `rank 1`) that looks at the rendezvous file `sleep_s` seconds after it began:
    (accepted?, its nonce, its process start, the time of the probe)."""
    import sys
    code = ("import ctypes, time, sys\n"
            "time.sleep(%r)\n"
            "R = ctypes.CDLL(%r)\n"
            "R.kfx_rccl_process_start.restype = ctypes.c_longlong\n"
            "R.kfx_rccl_launch_nonce.restype = ctypes.c_ulonglong\n"
            "print(R.kfx_rccl_rendezvous_probe(%r.encode()), R.kfx_rccl_launch_nonce(), R.kfx_rccl_process_start(), time.time())\n") % (sleep_s, lib, str(path))
    return subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_rccl_rendezvous_accepts_a_rank_that_arrives_seconds_later(tmp_path):
    """Round-3 advice: rank 0 publishes the id, rank 1 -- slower GPU init, import under 8-rank contention -- gets to the
    rendezvous five seconds later and must still accept the file.  (The old test of "not older than this process" used the
    st_mtime of /proc/self, which procfs stamps at the first lookup, i.e. when the late rank arrives: it rejected the valid file
    until the timeout.)  Host-only through the library's probe hook: a fresh process whose first look at procfs comes after the
    sleep; a file from long before the process began is still refused."""
    import time
    lib, R = _rccl_lib()
    path = tmp_path / "id"
    env = {k: v for k, v in os.environ.items() if k not in _RDV_ENV}
    env["KFX_RUN_ID"] = "late-rank"
    # the nonce of this launch's ranks = children of THIS process: ask one
    first = _probe_child(lib, path, env)
    nonce = int(first.communicate(timeout=60)[0].split()[1])
    t_launch = time.time()
    proc = _probe_child(lib, path, env, sleep_s=5.0)    # rank 1 is busy importing / initialising its GPU
    path.write_bytes(_rdv_payload(nonce))               # rank 0 publishes right after the launch
    out = proc.communicate(timeout=60)[0].split()
    assert proc.returncode == 0 and int(out[0]) == 1, out
    assert abs(int(out[2]) - t_launch) <= 2, "process start from /proc/self/stat: %s vs %.1f" % (out[2], t_launch)
    assert float(out[3]) - int(out[2]) >= 4.5           # it did arrive five seconds after it began
    # ... and a file from an earlier run (an hour before this process began) is still refused
    os.utime(path, (time.time() - 3600, time.time() - 3600))
    probe = _probe_child(lib, path, env)
    so, se = probe.communicate(timeout=60)
    assert probe.returncode == 0 and int(so.split()[0]) == 0, so + se


def test_rccl_rendezvous_refuses_the_file_of_a_crashed_run_of_the_same_port(tmp_path):
    """Round-4 advice: torchrun's default MASTER_PORT makes the environment part of the nonce static across relaunches, and a
    run that crashed inside ncclCommInitRank leaves its file behind.  (a) With the environment's nonce alone (KFX_RDV_PARENT=0) a
    file 30 s older than the new rank is outside the launch skew (10 s) and refused, one 3 s older is accepted, and KFX_RDV_SLACK_S
    moves the limit.  (b) By default the nonce also carries the launcher (the ranks' parent process): the same environment under
    another parent gives another nonce, so the crashed run's file is refused whatever its age."""
    import sys
    import time
    lib, R = _rccl_lib()
    path = tmp_path / "id"
    env = {k: v for k, v in os.environ.items() if k not in _RDV_ENV and k != "KFX_RUN_ID"}
    env["MASTER_PORT"] = "29500"
    # (a) environment-only nonce: the age test is what protects
    env_a = dict(env, KFX_RDV_PARENT="0")
    nonce_a = int(_probe_child(lib, path, env_a).communicate(timeout=60)[0].split()[1])
    path.write_bytes(_rdv_payload(nonce_a))
    for age, slack, want in ((30, None, 0), (60, None, 0), (3, None, 1), (30, "120", 1), (3, "0", 0)):
        os.utime(path, (time.time() - age, time.time() - age))
        e = dict(env_a) if slack is None else dict(env_a, KFX_RDV_SLACK_S=slack)
        so, se = _probe_child(lib, path, e).communicate(timeout=60)
        assert int(so.split()[0]) == want, (age, slack, so, se)
    # (b) default: a relaunch has another launcher -> another nonce, even for a fresh file
    nonce_here = int(_probe_child(lib, path, env).communicate(timeout=60)[0].split()[1])
    via = ("import subprocess, sys\n"
           "sys.exit(subprocess.call([sys.executable, '-c', sys.argv[1]]))\n")
    inner = ("import ctypes\n"
             "R = ctypes.CDLL(%r)\n"
             "R.kfx_rccl_launch_nonce.restype = ctypes.c_ulonglong\n"
             "print(R.kfx_rccl_rendezvous_probe(%r.encode()), R.kfx_rccl_launch_nonce())\n") % (lib, str(path))
    path.write_bytes(_rdv_payload(nonce_here))   # the file of the run whose launcher is THIS process, fresh
    assert int(_probe_child(lib, path, env).communicate(timeout=60)[0].split()[0]) == 1
    out = subprocess.run([sys.executable, "-c", via, inner], env=env, capture_output=True, text=True, timeout=60)
    got = out.stdout.split()
    assert out.returncode == 0 and int(got[0]) == 0 and int(got[1]) != nonce_here, out.stdout + out.stderr
    assert nonce_a != nonce_here


def test_rccl_rendezvous_named_launches_agree_across_parents_and_a_foreign_nonce_is_reported(tmp_path):
    """Round-5 advice: ranks that do not share a parent (one wrapper shell per rank, a step daemon per task, one agent per node)
    derived different nonces and ignored rank 0's valid file until the timeout, silently.  (a) A launch that names itself
    (KFX_RUN_ID, or a TORCHELASTIC_RUN_ID other than torchrun's static "none") leaves the parent out of the nonce: a rank started
    through another parent accepts the file; KFX_RDV_PARENT=1 puts the parent back.  (b) When the wait times out on a fresh,
    well-formed file whose only mismatch is the nonce, the error says so."""
    import sys
    lib, R = _rccl_lib()
    path = tmp_path / "id"
    base = {k: v for k, v in os.environ.items() if k not in _RDV_ENV and k != "KFX_RUN_ID"}
    via = ("import subprocess, sys\n"
           "sys.exit(subprocess.call([sys.executable, '-c', sys.argv[1]]))\n")
    inner = ("import ctypes\n"
             "R = ctypes.CDLL(%r)\n"
             "R.kfx_rccl_launch_nonce.restype = ctypes.c_ulonglong\n"
             "print(R.kfx_rccl_rendezvous_probe(%r.encode()), R.kfx_rccl_launch_nonce())\n") % (lib, str(path))

    def other_parent(env):
        out = subprocess.run([sys.executable, "-c", via, inner], env=env, capture_output=True, text=True, timeout=60)
        assert out.returncode == 0, out.stdout + out.stderr
        return [int(v) for v in out.stdout.split()]
    for env in (dict(base, KFX_RUN_ID="job-17"), dict(base, TORCHELASTIC_RUN_ID="elastic-17", MASTER_PORT="29500")):
        nonce = int(_probe_child(lib, path, env).communicate(timeout=60)[0].split()[1])
        path.write_bytes(_rdv_payload(nonce))
        assert other_parent(env) == [1, nonce], env                       # another parent, same name: same nonce, file accepted
        got = other_parent(dict(env, KFX_RDV_PARENT="1"))                  # the parent forced back in: another nonce
        assert got[0] == 0 and got[1] != nonce
    # torchrun's default run id is the static string "none": the launcher stays part of the nonce
    env = dict(base, TORCHELASTIC_RUN_ID="none", MASTER_PORT="29500")
    nonce = int(_probe_child(lib, path, env).communicate(timeout=60)[0].split()[1])
    path.write_bytes(_rdv_payload(nonce))
    got = other_parent(env)
    assert got[0] == 0 and got[1] != nonce
    # (b) the timeout names the reason
    code = ("import ctypes\n"
            "R = ctypes.CDLL(%r)\n"
            "comm = ctypes.create_string_buffer(256)\n"
            "R.kfx_comm_create_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]\n"
            "print(R.kfx_comm_create_rccl(comm, 1, 2, %r.encode(), 1))\n") % (lib, str(path))
    path.write_bytes(_rdv_payload(nonce ^ 2))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["-4"] and "launch nonce differs" in out.stderr and "KFX_RUN_ID" in out.stderr, out.stdout + out.stderr
    path.unlink()
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["-4"] and "launch nonce differs" not in out.stderr and "timed out waiting" in out.stderr, out.stdout + out.stderr


def test_frame_object_validates_its_views_without_a_gpu():
    """kfx_frame_create (include/kfx.h) checks the caller's views before any HIP call: with no timing slots it needs no device, so
    the argument errors -- and a well-formed create / destroy -- are checked here."""
    L = _lib.load()
    cfg = _lib.KfxFrameConfig()
    h = C.c_void_p()
    assert L.kfx_frame_create(C.byref(h), None) == -1 and L.kfx_frame_create(None, C.byref(cfg)) == -1
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == -1                     # null volume
    fake = 0x10000
    cfg.vol = _lib.KfxVolume(64 * 8, fake, 64, 64, 64 * 8 * 64, 64)
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == -2                     # image views missing
    w, hh = 40, 30
    f1, f4 = _lib.KfxImage(w * 4, fake, w, hh), _lib.KfxImage(w * 16, fake, w, hh)
    cfg.raw, cfg.filtered, cfg.vbo, cfg.normals, cfg.ray_depth, cfg.ray_norm, cfg.ray_img = f1, f1, f4, f4, f1, f4, f1
    cfg.timing_slots = -1
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == -4                     # KFX_E_RANGE
    cfg.timing_slots = 0
    cfg.vbo = _lib.KfxImage((w + 1) * 16, fake, w + 1, hh)
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == -2                     # the preprocess images differ in size
    cfg.vbo = _lib.KfxImage(w * 8, fake, w, hh)
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == -2                     # pitch below a row of float4
    cfg.vbo = f4
    assert L.kfx_frame_create(C.byref(h), C.byref(cfg)) == 0 and h.value
    assert L.kfx_frame_get_track(h) == 0 and L.kfx_frame_count(h) == 0 and L.kfx_frame_summary(h) is None
    assert L.kfx_frame_set_timing(h, 16) == -4 and L.kfx_frame_set_timing(h, 6) == 0
    out = (C.c_float * 5)()
    assert L.kfx_frame_timings(h, 0, 1, out) == -4                                # created without timing slots
    assert L.kfx_frame_destroy(h) == 0


def test_composite_strip_layout_and_argument_checks_without_a_gpu():
    """kfx_composite_strip_pixels (the direct-send merge's strip size: whole 256-byte planes, the last strip padded) and the
    argument validation of the three strip kernels' entry points (no launch is reached)."""
    L = _lib.load()
    S = L.kfx_composite_strip_pixels
    assert S(640, 480, 8) == 38400 and S(640, 480, 1) == 307200 and S(161, 97, 3) == 5248 and S(1, 1, 4) == 64
    assert S(640, 480, 0) == 0 and S(0, 0, 2) == 0
    for w, h, n in ((640, 480, 8), (161, 97, 3), (320, 240, 7)):
        s = S(w, h, n)
        assert s % 64 == 0 and s * n >= w * h > (s - 64) * n
    assert L.kfx_composite_strips_merge(None, None, 64, 0, 2, None) == -1                     # KFX_E_NULL
    buf = C.create_string_buffer(4096)
    assert L.kfx_composite_strips_merge(buf, buf, 64, 0, 0, None) == -4                       # KFX_E_RANGE: world
    assert L.kfx_composite_strips_merge(buf, buf, 64, 100, 2, None) == -2                     # KFX_E_SHAPE: rank stride smaller than a strip
    assert L.kfx_composite_strips_pack(None, None, None, buf, 0, 2, None) == -1
    assert L.kfx_composite_strips_unpack(None, None, None, buf, 0, 2, None) == -1
    assert L.kfx_slab_composite_direct(None, None, None, None, None, None) == -1
    assert L.kfx_slab_composite_direct_scratch_bytes(640, 480, 8) == (2 * 8 + 1) * 5 * 38400 * 4


def test_slab_frame_and_tiled_march_validate_their_arguments_without_a_gpu():
    """kfx_slab_frame_create / kfx_slab_raycast_exact_tiled (include/kfx_slab.h) check views, layout and policies before any HIP call;
    the in-process and loop-back transports are plain host objects; the ctypes mirrors have the C structs' sizes."""
    from kangaroo_amd import slab
    L = slab._L()
    # sizes of the C structs (gcc) against the ctypes mirrors
    src = os.path.join(T.ROOT, "include")
    code = '#include <cstdio>\n#include "kfx_slab.h"\nint main(){ printf("%zu %zu %zu\\n", sizeof(kfx_slab_frame_config), sizeof(kfx_comm), sizeof(kfx_slab_layout)); }\n'
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        open(os.path.join(td, "s.cpp"), "w").write(code)
        subprocess.check_call(["g++", "-I", src, os.path.join(td, "s.cpp"), "-o", os.path.join(td, "s")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(td, "s")]).split()]
    assert sizes == [C.sizeof(slab.KfxSlabFrameConfig), C.sizeof(slab.KfxComm), C.sizeof(slab.KfxSlabLayout)]
    comms = slab.Comm.threads(3)
    assert [c.rank for c in comms] == [0, 1, 2] and all(c.world == 3 for c in comms) and bool(comms[1].c.exchange_v)
    comms[0].destroy()
    lb = slab.Comm.loopback(3, 8)
    assert (lb.rank, lb.world) == (3, 8) and bool(lb.c.all_to_all) and bool(lb.c.exchange_v)
    assert L.kfx_comm_create_loopback(None, 0, 1) == -1 and L.kfx_comm_create_loopback(C.byref(slab.KfxComm()), 2, 2) == -4
    # the frame object
    h = C.c_void_p()
    cfg = slab.KfxSlabFrameConfig()
    assert L.kfx_slab_frame_create(C.byref(h), None, lb.ref()) == -1 and L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), None) == -1
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -1            # null volume
    fake, w, hh = 0x10000, 40, 30
    lay = slab.layout(64, 2.0, 4.0, 3, 8, 2)
    cfg.layout = lay
    cfg.local = _lib.KfxVolume(64 * 8, fake, 64, 64, 64 * 8 * 64, lay.s1 - lay.s0)
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -2            # image views missing
    f1, f4 = _lib.KfxImage(w * 4, fake, w, hh), _lib.KfxImage(w * 16, fake, w, hh)
    cfg.raw, cfg.filtered, cfg.vbo, cfg.normals, cfg.ray_depth, cfg.ray_norm, cfg.ray_img = f1, f1, f4, f4, f1, f4, f1
    other = slab.Comm.loopback(2, 8)
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), other.ref()) == -2        # the communicator's rank is not the layout's
    cfg.local.d = 5
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -2            # plane count does not match the layout
    cfg.local.d = lay.s1 - lay.s0
    for field, bad in (("halo", 2), ("raycast", 7), ("merge", -1), ("inputs", 3), ("tiles", 65)):
        setattr(cfg, field, bad)
        assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -4, field
        setattr(cfg, field, 0)
    cfg.overlap, cfg.raycast = 1, slab.RAYCAST["exact"]                                   # an overlapped merge belongs to the composite ...
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -4 and b"overlap needs" in L.kfx_last_error_string()
    cfg.raycast, cfg.halo = slab.RAYCAST["composite"], slab.HALO["exchange"]             # ... with nothing else communicating
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -4
    cfg.timing_slots, cfg.overlap = -1, 0
    assert L.kfx_slab_frame_create(C.byref(h), C.byref(cfg), lb.ref()) == -4
    assert L.kfx_slab_frame_step(None, None, None, None, 0, None) == -1 and L.kfx_slab_frame_wait(None, None) == -1
    assert L.kfx_slab_frame_count(None) == 0 and L.kfx_slab_frame_destroy(None) == 0
    # the tiled hand-over: scratch size grows with the tile count only by the padding of the last tile, arguments are checked
    one, four = L.kfx_slab_exact_tiled_scratch_bytes(640, 480, 1, 1), L.kfx_slab_exact_tiled_scratch_bytes(640, 480, 4, 1)
    assert one == four == (4 + 4 + 4 + 4 + 1 + 6) * 640 * 480 * 4 + 256 and L.kfx_slab_exact_tiled_scratch_bytes(640, 487, 8, 1) >= one
    eight = L.kfx_slab_exact_tiled_scratch_bytes(640, 480, 4, 8)   # + the strips of the final exchange: [8][6][S] twice over, [6][S]
    assert eight == (4 + 4 + 4 + 4 + 1) * 640 * 480 * 4 + (2 * 8 + 1) * 6 * 38400 * 4 + 256
    assert L.kfx_slab_raycast_exact_tiled(None, None, None, None, None, None, None, None, 0.4, 4.0, 0.01, 1, 4, None, None, None, None) == -1


def test_depth_pyramid_entry_point_validates_its_levels_without_a_gpu():
    """kfx_depth_pyramid_vbo_normals_f32 (the pyramid + maps in one launch): level count, per-level sizes and alignment are checked
    before any launch -- fake non-null pointers are enough to get the status codes."""
    L = _lib.load()
    def img(w, h, elem, ptr=0x10000):
        return _lib.KfxImage(w * elem, ptr, w, h)
    K = (C.c_float * 16)(*([500.0, 500.0, 320.0, 240.0] * 4))
    def call(sizes, levels, vbo_sizes=None, ptr=0x10000):
        n = len(sizes)
        d = (_lib.KfxImage * n)(*[img(w, h, 4, ptr) for w, h in sizes])
        v = (_lib.KfxImage * n)(*[img(w, h, 16, ptr) for w, h in (vbo_sizes or sizes)])
        m = (_lib.KfxImage * n)(*[img(w, h, 16, ptr) for w, h in (vbo_sizes or sizes)])
        return L.kfx_depth_pyramid_vbo_normals_f32(d, v, m, K, levels, 1.0, None)
    four = [(64, 48), (32, 24), (16, 12), (8, 6)]
    assert call(four, 0) == -4 and call(four + [(4, 3)], 5) == -4                    # KFX_E_RANGE: 1 .. 4 levels
    assert call([(64, 48), (40, 24)], 2) == -2                                        # KFX_E_SHAPE: a level larger than half the one before
    assert call([(64, 48), (32, 24)], 2, vbo_sizes=[(64, 48), (32, 20)]) == -2        # maps of another size than their depth level
    assert call([(64, 48), (0, 24)], 2) == -2                                         # an empty level
    assert call(four, 4, ptr=0x10004) == -3                                           # KFX_E_ALIGN: float4 maps at a 4-byte boundary
    assert L.kfx_depth_pyramid_vbo_normals_f32(None, None, None, K, 1, 1.0, None) == -1
