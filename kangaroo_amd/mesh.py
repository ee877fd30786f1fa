"""Mesh extraction: roo::SaveMesh (reference include/kangaroo/MarchingCubes.h:205-262) with the volume left in HBM.

The reference copies the volume to the host, marches the (w-1)(h-1)(d-1) cubes one by one and hands the lists to
Assimp's PLY exporter.  Here: kfx_mc_count (triangles per cube, in the reference's emission order) -> exclusive
compaction of the active cubes and prefix sum over them (torch.nonzero / torch.cumsum on the device) -> kfx_mc_emit (one
thread per active cube: vertices, normals, grey colours into their slots).  The
arrays equal the host algorithm's element for element (tests compare with the oracle); only the finished arrays
cross PCIe.

Case tables: kangaroo_amd/csrc/mc_tables.inc, derived by scripts/gen_mc_tables.py.  Their boundary loops and
winding agree with the classic tables the reference uses in all 256 cases; 158 cases split a polygon along a
different interior diagonal (an equally valid triangulation of the same loop), so meshes are the same surface
patch by patch but not triangle-for-triangle identical to the reference's.

PLY: Assimp (an external library of the reference, absent here) writes the reference's file; this writer emits the
same element set -- per-vertex position, normal and, with a colour volume, RGBA floats; one triangle per three
consecutive vertices -- as a standard ascii or binary_little_endian PLY.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .roo import _stream


def ExtractMesh(vol, colorVol=None, stream=None):
    """Returns (verts, norms, colors): float32 device tensors of shape (3T, 3), (3T, 3) and (3T, 4) or None."""
    L = _lib.load()
    cx, cy, cz = vol.w - 1, vol.h - 1, vol.d - 1
    dev = vol.storage.device
    counts = torch.empty(cx * cy * cz, dtype=torch.uint8, device=dev)
    _lib.check(L.kfx_mc_count(vol.ref(), C.c_void_p(counts.data_ptr()), _stream(stream)))
    active = torch.nonzero(counts).reshape(-1)                        # cubes with triangles, ascending = emission order
    ca = counts[active].to(torch.int64)                               # the scan only needs them: empty cubes add nothing
    incl = torch.cumsum(ca, 0)
    ntri = int(incl[-1].item()) if incl.numel() else 0
    if ntri >= 2 ** 32 // 3:
        raise ValueError("mesh too large for 32-bit vertex offsets")
    tri_offset = (incl - ca).to(torch.int32)                          # exclusive prefix sum at those cubes
    verts = torch.empty((3 * ntri, 3), dtype=torch.float32, device=dev)
    norms = torch.empty((3 * ntri, 3), dtype=torch.float32, device=dev)
    has_color = colorVol is not None and min(colorVol.w, colorVol.h, colorVol.d) >= 8
    colors = torch.empty((3 * ntri, 4), dtype=torch.float32, device=dev) if has_color else None
    if ntri:
        _lib.check(L.kfx_mc_emit(vol.ref(), colorVol.ref() if has_color else None, C.c_void_p(active.data_ptr()),
                                 C.c_void_p(tri_offset.data_ptr()), int(active.numel()), C.c_void_p(verts.data_ptr()),
                                 C.c_void_p(norms.data_ptr()), C.c_void_p(colors.data_ptr()) if has_color else None, _stream(stream)))
    return verts, norms, colors


def write_ply(path, verts, norms, colors=None, binary=True):
    verts, norms = np.asarray(verts, np.float32), np.asarray(norms, np.float32)
    n = len(verts)
    cols = [verts, norms] + ([np.asarray(colors, np.float32)] if colors is not None else [])
    table = np.concatenate(cols, axis=1)
    names = ["x", "y", "z", "nx", "ny", "nz"] + (["red", "green", "blue", "alpha"] if colors is not None else [])
    head = ["ply", "format %s 1.0" % ("binary_little_endian" if binary else "ascii"),
            "comment kangaroo_amd marching cubes", "element vertex %d" % n]
    head += ["property float %s" % nm for nm in names]
    head += ["element face %d" % (n // 3), "property list uchar uint vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        if binary:
            f.write(table.astype("<f4").tobytes())
            faces = np.empty(n // 3, dtype=[("k", "u1"), ("i", "<u4", 3)])
            faces["k"] = 3
            faces["i"] = np.arange(n, dtype=np.uint32).reshape(-1, 3)
            f.write(faces.tobytes())
        else:
            for row in table:
                f.write((" ".join("%.9g" % v for v in row) + "\n").encode())
            for i in range(0, n, 3):
                f.write(("3 %d %d %d\n" % (i, i + 1, i + 2)).encode())


def SaveMesh(filename, vol, colorVol=None, binary=True):
    """SaveMesh(filename, vol[, volColor]) (MarchingCubes.h:246-262): writes filename + ".ply"; returns the triangle count."""
    verts, norms, colors = ExtractMesh(vol, colorVol)
    torch.cuda.synchronize() if verts.is_cuda else None
    write_ply(filename + ".ply", verts.cpu().numpy(), norms.cpu().numpy(), None if colors is None else colors.cpu().numpy(), binary)
    return len(verts) // 3
