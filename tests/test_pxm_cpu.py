"""SURVEY 8(f) row f-4, persistence: SavePXM / LoadPXM of BoundedVolume.  The writer must produce the bytes the
reference's own SavePXM produces (oracle/_ref drives the reference's stream writer), the reader must parse
them back, and the C++ header (include/kangaroo/extra/SavePPM.h) is exercised by apps/roo_api_test on the GPU."""
import ctypes as C
import os

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle, scenes
from kangaroo_amd import pxm, roo

REF_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")


def host_volume(w, h, d, bmin, bmax, kind="f32"):
    return roo.BoundedVolume(w, h, d, bmin, bmax, device="cpu", kind=kind)


def filled(kind, dims=(12, 7, 5), bmin=(-1.25, -0.333333343, 2.0), bmax=(1.0, 0.1, 4.000001)):
    rng = np.random.default_rng(7)
    vol = host_volume(dims[0], dims[1], dims[2], bmin, bmax, kind)
    ch = 1 if kind == "c32" else 2
    data = rng.standard_normal((dims[2], dims[1], dims[0], ch)).astype(np.float16 if kind == "f16" else np.float32)
    data[0, 0, 0, 0] = np.nan
    vol.MemcpyFromHost(data)
    return vol, data


@pytest.mark.parametrize("kind", ["f32", "f16", "c32"])
def test_pxm_round_trip(tmp_path, kind):
    vol, data = filled(kind)
    path = str(tmp_path / "save.vol")
    pxm.SavePXM(path, vol)
    back = pxm.LoadPXM(path, host_volume, kind=kind)
    assert back is not None and (back.w, back.h, back.d) == (vol.w, vol.h, vol.d)
    assert T.nan_equal(back.MemcpyToHost(), data)
    # the text header keeps 6 significant digits of the box (operator<< float)
    assert np.allclose(back.boxmin, vol.boxmin, rtol=1e-5) and np.allclose(back.boxmax, vol.boxmax, rtol=1e-5)
    head = open(path, "rb").read(80).split(b"\n")
    assert head[0] == b"-1.25 -0.333333 2" and head[1] == b"1 0.1 4" and head[2] == b"P5" and head[3] == b"12 7 5" and head[4] == b"255"
    # truncated payload / header -> load fails like the reference (returns false)
    raw = open(path, "rb").read()
    open(path, "wb").write(raw[:-5])
    assert pxm.LoadPXM(path, host_volume, kind=kind) is None
    open(path, "wb").write(raw[:20])
    assert pxm.LoadPXM(path, host_volume, kind=kind) is None


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("kind,elem", [("f32", 8), ("c32", 4)])
def test_pxm_bytes_equal_the_reference_writer(tmp_path, kind, elem):
    R = C.CDLL(REF_SO)
    R.ref_save_pxm.argtypes = [C.c_char_p, C.c_void_p, C.c_int]
    vol, data = filled(kind)
    ovol = oracle.Volume(vol.w, vol.h, vol.d, vol.boxmin, vol.boxmax, pitch_bytes=vol.w * elem + 24, elem_floats=elem // 4)
    ovol.data[...] = data
    ref_path, my_path = str(tmp_path / "ref.vol"), str(tmp_path / "mine.vol")
    assert R.ref_save_pxm(ref_path.encode(), ovol.ref(), elem) == 0
    pxm.SavePXM(my_path, vol)
    assert open(ref_path, "rb").read() == open(my_path, "rb").read()
    back = pxm.LoadPXM(ref_path, host_volume, kind=kind)
    assert T.nan_equal(back.MemcpyToHost(), data)
