#!/usr/bin/env python3
"""Derives the marching-cubes case tables used by kangaroo_amd/csrc/mesh.hip and writes mc_tables.inc.

Conventions (the classic numbering the reference's extraction also uses, MarchingCubes.h:58-66):
  corners 0..7 at (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1);
  edges 0..11 = corner pairs 0-1 1-2 2-3 3-0 4-5 5-6 6-7 7-4 0-4 1-5 2-6 3-7;
  case index bit i set <=> value at corner i <= iso.

Derivation (not a transcription of any published table):
  1. an edge carries a vertex iff its end corners differ;
  2. on each of the 6 faces the edge vertices are joined by segments: 2 vertices -> 1 segment; 4 vertices (the
     two inside corners are diagonal) -> 2 segments, each cutting off one INSIDE corner.  The rule depends only on
     the face's own corner signs, so the two cubes sharing a face always agree: the surface has no cracks (the
     classic table is complement-symmetric instead and cracks on such faces);
  3. segments chain into closed loops; every loop is oriented so that its normal (Newell) points from the outside
     corners it separates towards the inside ones (the winding of the classic table, so that a consumer sees the
     same front faces), rotated to start at its smallest edge index, and fan-triangulated from that vertex.
Output: per case the number of triangles and up to 15 edge indices (-1 padded), plus the 12-bit edge mask.
"""
import itertools
import os
import sys

CORNER = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)]
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]
# faces as corner cycles (consecutive corners share a cube edge)
FACES = [(0, 1, 2, 3), (4, 5, 6, 7), (0, 1, 5, 4), (1, 2, 6, 5), (2, 3, 7, 6), (3, 0, 4, 7)]
EDGE_OF = {frozenset(e): i for i, e in enumerate(EDGE)}


def edge_mid(e):
    a, b = (CORNER[c] for c in EDGE[e])
    return tuple((a[k] + b[k]) / 2.0 for k in range(3))


def case_loops(case):
    inside = [(case >> i) & 1 for i in range(8)]
    adj = {}

    def link(e0, e1):
        adj.setdefault(e0, []).append(e1)
        adj.setdefault(e1, []).append(e0)

    for f in FACES:
        cut = []   # (edge index, position k in the cycle: edge between f[k] and f[k+1])
        for k in range(4):
            a, b = f[k], f[(k + 1) % 4]
            if inside[a] != inside[b]:
                cut.append((EDGE_OF[frozenset((a, b))], k))
        if len(cut) == 2:
            link(cut[0][0], cut[1][0])
        elif len(cut) == 4:
            # corners alternate; cut off each inside corner f[k]: join the two face edges meeting at it
            for k in range(4):
                if inside[f[k]]:
                    e_prev = EDGE_OF[frozenset((f[(k - 1) % 4], f[k]))]
                    e_next = EDGE_OF[frozenset((f[k], f[(k + 1) % 4]))]
                    link(e_prev, e_next)
    loops, seen = [], set()
    for start in sorted(adj):
        if start in seen:
            continue
        loop, prev, cur = [start], None, start
        seen.add(start)
        while True:
            nxt = [n for n in adj[cur] if n != prev] or adj[cur]
            # a vertex has exactly two neighbours; walk away from where we came from
            n = nxt[0] if (prev is None or len(nxt) == 1) else nxt[0]
            if prev is not None and adj[cur].count(prev) == 2:   # 2-cycle cannot occur on a cube; guard anyway
                n = prev
            if n == start:
                break
            loop.append(n)
            seen.add(n)
            prev, cur = cur, n
        loops.append(loop)
    # orientation: the Newell normal of a loop points from the outside corners it touches to the inside ones
    oriented = []
    for loop in loops:
        pts = [edge_mid(e) for e in loop]
        n = [0.0, 0.0, 0.0]
        for i in range(len(pts)):
            p, q = pts[i], pts[(i + 1) % len(pts)]
            n[0] += (p[1] - q[1]) * (p[2] + q[2])
            n[1] += (p[2] - q[2]) * (p[0] + q[0])
            n[2] += (p[0] - q[0]) * (p[1] + q[1])
        # direction reference for THIS loop: from the inside corners it touches to the outside corners it touches
        tin = [CORNER[c] for e in loop for c in EDGE[e] if inside[c]]
        tout = [CORNER[c] for e in loop for c in EDGE[e] if not inside[c]]
        d = [sum(p[k] for p in tout) / len(tout) - sum(p[k] for p in tin) / len(tin) for k in range(3)]
        if sum(n[k] * d[k] for k in range(3)) > 0:
            loop = loop[::-1]
        m = loop.index(min(loop))
        oriented.append(loop[m:] + loop[:m])
    oriented.sort(key=lambda l: l[0])
    return oriented


def build():
    tri, ntri, mask = [], [], []
    for case in range(256):
        if case in (0, 255):
            tri.append([-1] * 15)
            ntri.append(0)
            mask.append(0)
            continue
        loops = case_loops(case)
        row, m = [], 0
        for loop in loops:
            for e in loop:
                m |= 1 << e
            for i in range(1, len(loop) - 1):
                row += [loop[0], loop[i], loop[i + 1]]
        assert len(row) <= 15, (case, row)
        ntri.append(len(row) // 3)
        tri.append(row + [-1] * (15 - len(row)))
        mask.append(m)
    return ntri, tri, mask


def render():
    ntri, tri, mask = build()
    out = ["// mc_tables.inc -- GENERATED by scripts/gen_mc_tables.py (do not edit; tests/test_mesh_cpu.py regenerates and compares).",
           "// Marching-cubes case tables derived from the cube's topology: see the generator for the rule set.",
           "static const unsigned char MC_NUM_TRIS[256] = {"]
    for i in range(0, 256, 32):
        out.append("    " + ", ".join(str(v) for v in ntri[i:i + 32]) + ",")
    out.append("};")
    out.append("static const unsigned short MC_EDGE_MASK[256] = {")
    for i in range(0, 256, 16):
        out.append("    " + ", ".join("0x%03x" % v for v in mask[i:i + 16]) + ",")
    out.append("};")
    out.append("static const signed char MC_TRIS[256][15] = {")
    for c in range(256):
        out.append("    {" + ", ".join("%2d" % v for v in tri[c]) + "},")
    out.append("};")
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kangaroo_amd", "csrc", "mc_tables.inc")
    text = render()
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        sys.exit(0 if open(path).read() == text else 1)
    open(path, "w").write(text)
    ntri, tri, mask = build()
    print("wrote %s: %d cases with triangles, max %d triangles, %d total" % (path, sum(1 for n in ntri if n), max(ntri), sum(ntri)))
