"""SURVEY 8(f) row f-4, mesh extraction: the generated case tables (scripts/gen_mc_tables.py) against the
reference's own tables (topology, via oracle/_ref), the oracle's marching cubes on an analytic sphere, and the
committed mc_tables.inc against a fresh run of its generator."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import kfx_testlib as T
from kfx_testlib import oracle

sys.path.insert(0, os.path.join(T.ROOT, "scripts"))
import gen_mc_tables as G  # noqa: E402

REF_SO = os.path.join(T.ROOT, "oracle", "_ref", "libkfx_refhdr.so")


def tables():
    ntri, tri, mask = G.build()
    return np.array(ntri, np.uint8), np.array(mask, np.uint16), np.array(tri, np.int8)


def test_committed_tables_are_the_generators_output():
    assert open(os.path.join(T.ROOT, "kangaroo_amd", "csrc", "mc_tables.inc")).read() == G.render()


def test_tables_are_consistent():
    """Every case: the edge mask is exactly the set of sign-changing edges, every triangle uses masked edges,
    every loop edge is shared by <= 2 triangles, and complementary cases have the same loops reversed or equal."""
    ntri, mask, tri = tables()
    for c in range(256):
        expect = 0
        for e, (a, b) in enumerate(G.EDGE):
            if ((c >> a) & 1) != ((c >> b) & 1):
                expect |= 1 << e
        assert mask[c] == expect
        used = [int(v) for v in tri[c] if v >= 0]
        assert len(used) == 3 * ntri[c] and all(expect & (1 << e) for e in used)
        assert set(used) == {e for e in range(12) if expect & (1 << e)}
    assert ntri.max() == 5 and int(ntri.sum()) == 820


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built (needs /root/reference)")
def test_tables_have_the_reference_tables_topology():
    """Against MarchingCubesTables.h: identical edge masks and triangle counts, and -- as oriented boundary loops
    -- identical surfaces on the cube's faces in all 256 cases (so adjacent cubes meet the same way and front
    faces agree); only interior diagonals of polygons with more than three corners may differ."""
    from collections import Counter
    R = C.CDLL(REF_SO)
    ef, tr = (C.c_int * 256)(), (C.c_int * (256 * 16))()
    R.ref_mc_tables(ef, tr)
    rt = np.array(list(tr)).reshape(256, 16)
    ntri, mask, tri = tables()

    def boundary(t):
        cnt = Counter()
        for i in range(0, len(t), 3):
            a, b, c = t[i:i + 3]
            for e in ((a, b), (b, c), (c, a)):
                cnt[e] += 1
        return frozenset(e for e in cnt if cnt.get((e[1], e[0]), 0) == 0)

    same_tris = 0
    for c in range(256):
        r = [int(v) for v in rt[c] if v >= 0]
        m = [int(v) for v in tri[c] if v >= 0]
        assert mask[c] == ef[c] and len(r) == len(m)
        assert boundary(r) == boundary(m), c
        canon = lambda t: frozenset(tuple(t[i:i + 3][j:] + t[i:i + 3][:j]) for i in range(0, len(t), 3)
                                    for j in [t[i:i + 3].index(min(t[i:i + 3]))])
        same_tris += canon(r) == canon(m)
    assert same_tris >= 98


def sphere_volume(N, r=0.7):
    vol = oracle.Volume(N, N, N, (-1, -1, -1), (1, 1, 1))
    oracle.sdf_sphere(vol, (0.05, -0.02, 0.03), r)
    return vol


def test_oracle_mesh_of_a_sphere():
    """Watertight, outward... the reference's winding (normals from the gradient point outward), every vertex on the
    sphere to interpolation accuracy, area within 1 % of 4 pi r^2."""
    N, r = 40, 0.7
    vol = sphere_volume(N, r)
    ntri, mask, tri = tables()
    verts, norms, colors = oracle.marching_cubes(vol, None, ntri, mask, tri)
    assert colors is None and len(verts) % 3 == 0 and len(verts) > 3000
    c = np.array([0.05, -0.02, 0.03], np.float32)
    rad = np.linalg.norm(verts - c, axis=1)
    voxel = 2.0 / (N - 1)
    assert np.abs(rad - r).max() < 0.05 * voxel
    nl = np.linalg.norm(norms, axis=1)
    assert np.allclose(nl, 1.0, atol=1e-5)
    assert ((norms * (verts - c) / rad[:, None]).sum(1) > 0.95).all()          # gradient of the SDF points outward
    t = verts.reshape(-1, 3, 3).astype(np.float64)
    cross = np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    area = 0.5 * np.linalg.norm(cross, axis=1).sum()
    assert abs(area / (4 * np.pi * r * r) - 1) < 0.01
    # watertight: weld the copies of a vertex (the cubes sharing an edge interpolate it from opposite ends, so their
    # copies differ in the last bits); every directed edge then occurs once, its reverse once, and V - E + F = 2
    from collections import Counter
    import scipy.sparse as sp
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    flat = t.reshape(-1, 3)
    pairs = np.array(sorted(cKDTree(flat).query_pairs(2e-6)))
    g = sp.coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(len(flat),) * 2)
    nv, ids = connected_components(g, directed=False)
    cnt = Counter()
    for tri3 in ids.reshape(-1, 3):
        assert len(set(tri3)) == 3
        for a, b in ((0, 1), (1, 2), (2, 0)):
            cnt[(int(tri3[a]), int(tri3[b]))] += 1
    assert all(n == 1 for n in cnt.values())
    assert all(cnt.get((b, a), 0) == 1 for (a, b) in cnt)
    assert nv - len(cnt) // 2 + len(t) == 2
    # winding: the reference's tables wind triangles so that their geometric normal points INTO the surface (towards
    # sdf <= 0); the per-vertex normals carry the outward direction
    geo = cross / np.linalg.norm(cross, axis=1, keepdims=True)
    cen = t.mean(1) - c
    assert ((geo * cen).sum(1) < 0).mean() > 0.999


def test_oracle_mesh_skips_unobserved_cells_and_samples_colour():
    N = 24
    vol = sphere_volume(N, 0.6)
    vol.data[: N // 2, :, :, 0] = np.nan            # half of the volume never observed
    cvol = oracle.ColorVolume(N, N, N, (-1, -1, -1), (1, 1, 1))
    cvol.data[...] = 0.25
    ntri, mask, tri = tables()
    verts, norms, colors = oracle.marching_cubes(vol, cvol, ntri, mask, tri)
    assert len(verts) > 0 and (verts[:, 2] > -2.0 / (N - 1)).all()
    assert colors is not None and np.all(colors[:, :3] == 0.25) and np.all(colors[:, 3] == 1.0)
    small = oracle.ColorVolume(4, 4, 4, (-1, -1, -1), (1, 1, 1))   # !IsValid(): no colours sampled (MarchingCubes.h:134)
    v2, n2, c2 = oracle.marching_cubes(vol, small, ntri, mask, tri)
    assert np.array_equal(v2, verts) and not c2.any()
