"""Volume persistence: SavePXM / LoadPXM for BoundedVolume (reference include/kangaroo/extra/SavePPM.h:42-195).

File layout (what the reference's iostream code produces / parses):

    <bmin.x> <bmin.y> <bmin.z>\\n        bbox, operator<< float: 6 significant digits ("%g")   SavePPM.h:83-84
    <bmax.x> <bmax.y> <bmax.z>\\n
    P5\\n                                 ppm_type                                                SavePPM.h:49
    <w> <h> <d>\\n                                                                                SavePPM.h:50
    255\\n                                num_colors                                              SavePPM.h:51
    d*h rows of w*sizeof(T) raw bytes    x fastest, rows packed (the pitch is not stored)        SavePPM.h:52-56

T is the cell type: 8-byte SDF_t, 4-byte SDF_h or 4-byte float (colour).  The bbox survives only to the
6 digits the text header holds -- a property of the format, reproduced as is.  The device <-> host hop is
one strided copy; everything else is host file IO.
"""
import re

import numpy as np
import torch

_TOKEN = re.compile(rb"\s*(\S+)")


def _g(x):
    return "%g" % float(np.float32(x))   # iostream default float formatting


def SavePXM(filename, vol, ppm_type="P5", num_colors=255):
    """SavePXM(filename, BoundedVolume<T,TargetDevice>&, ppm_type, num_colors) (SavePPM.h:78-87)."""
    cells = vol.tensor().contiguous().cpu().numpy()        # (d, h, w, c) packed rows
    with open(filename, "wb") as f:
        f.write(("%s %s %s\n" % tuple(_g(v) for v in vol.boxmin)).encode())
        f.write(("%s %s %s\n" % tuple(_g(v) for v in vol.boxmax)).encode())
        f.write(("%s\n%d %d %d\n%d\n" % (ppm_type, vol.w, vol.h, vol.d, num_colors)).encode())
        f.write(cells.tobytes())


def _tokens(buf, pos, n):
    out = []
    for _ in range(n):
        m = _TOKEN.match(buf, pos)
        if m is None:
            raise ValueError("truncated PXM header")
        out.append(m.group(1))
        pos = m.end()
    return out, pos


def LoadPXM(filename, make_volume, kind="f32"):
    """LoadPXM(filename, BoundedVolume<T,TargetDevice,Manage>&) (SavePPM.h:180-195): parse the header the way
    the reference's `>>` extractions do (whitespace-separated tokens, one ignored byte after the bbox and after
    num_colors), allocate a volume of the stored size through `make_volume(w, h, d, boxmin, boxmax, kind=kind)`
    (e.g. roo.BoundedVolume) and fill it.  Returns the volume, or None when the header or payload is short
    (the reference returns false)."""
    buf = open(filename, "rb").read()
    try:
        box, pos = _tokens(buf, 0, 6)
        box = [float(np.float32(float(t))) for t in box]
        pos += 1                                             # bFile.ignore(1, '\n')
        (ppm_type, w, h, d, num_colors), pos = _tokens(buf, pos, 5)
        w, h, d, num_colors = int(w), int(h), int(d), int(num_colors)
        pos += 1
    except (ValueError, IndexError):
        return None
    if not (w > 0 and h > 0 and d > 0):
        return None
    elem, dt, ch = {"f32": (8, np.float32, 2), "f16": (4, np.float16, 2), "c32": (4, np.float32, 1)}[kind]
    need = w * h * d * elem
    if len(buf) - pos < need:
        return None
    vol = make_volume(w, h, d, box[:3], box[3:], kind=kind)
    cells = np.frombuffer(buf, dt, count=w * h * d * ch, offset=pos).reshape(d, h, w, ch)
    vol.tensor().copy_(torch.from_numpy(cells.copy()))
    return vol
