#!/usr/bin/env python3
"""make_pmc_traffic.py FULL_SUMMARY ROOM_SUMMARY OUT.json -- condenses the FETCH_SIZE / WRITE_SIZE sections of two
scripts/gpu_profile.sh summaries (S_full and S_room runs of `python bench.py`) into the traffic file bench.py reads
(profiles/r04_pmc_traffic.json): per kernel family (2 x FETCH_SIZE + WRITE_SIZE) KiB -> bytes."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sections(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"== (FETCH_SIZE|WRITE_SIZE) per launch", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        if line.startswith("== "):
            cur = None
            continue
        if cur:
            m = re.match(r"(.+?)\s+n=\s*(\d+)\s+avg=\s*([0-9.]+)", line)
            if m:
                out[cur][m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return out


# (name prefixes: the summaries cut kernel names at 60 characters, and later template arguments -- waves per workgroup -- follow)
FUSE = {"fast": "k_sdf_fuse_tiled<true, 2, CellF32, 32, 4, 16, false, false", "fast_tracked": "k_sdf_fuse_tiled<true, 2, CellF32, 32, 4, 16, true, false"}
RAY = {"fast": "k_raycast_sdf<RayF32, false>", "fast_tracked": "k_raycast_sdf_classes<RayF32>"}


def main():
    full, room, dst = sections(sys.argv[1]), sections(sys.argv[2]), sys.argv[3]
    d = {"_note": "HBM-side traffic per call from rocprofv3 PMC passes of `python bench.py` (scripts/gpu_profile.sh: separate --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                  "runs; this file: scripts/make_pmc_traffic.py over the two summaries named on its command line: %s and %s). traffic = (2 x FETCH_SIZE + " % (sys.argv[1], sys.argv[2]) +
                  "WRITE_SIZE) KiB -> bytes: on gfx950 FETCH_SIZE tallies a 128-byte request at 64 bytes (MI355X_MICROARCH.md, HBM section: double it); WRITE_SIZE "
                  "equals the byte count of the known 1 GiB fills of the same passes (k_fill_sdf, k_rmw_*) and is used as is. The counters sit between the XCDs' L2 "
                  "and the fabric: reads served by the 256 MiB memory-side cache are counted like reads from HBM. SdfFuse in S_room is two launches per call "
                  "(added up here). RaycastSdf gathers 16-byte cell pairs and its requests beyond the L2 are whole 128-byte lines, so its traffic exceeds 8 B x "
                  "distinct cells by the cells of those lines no ray samples (1.3-1.8 x), not by re-reads.",
         "_commit": "the kernels whose source digests are recorded below (_kernel_source_id), as of %s and %s" % (os.path.dirname(sys.argv[1]), os.path.dirname(sys.argv[2]))}

    def entry(sec, name, launches=1):
        def find(table):
            hits = [v for k, v in table.items() if k.startswith(name)]
            return hits[0] if len(hits) == 1 else None
        f, w = find(sec["FETCH_SIZE"]), find(sec["WRITE_SIZE"])
        if not f or not w:
            return None
        fk, wk = f[1] * launches, w[1] * launches
        return {"fetch_kib": round(fk, 1), "write_kib": round(wk, 1), "traffic_bytes": int(round((2 * fk + wk) * 1024))}
    exact_full = [k for k in full["FETCH_SIZE"] if k.startswith("k_sdf_fuse_tiled<false")]
    exact_room = [k for k in room["FETCH_SIZE"] if k.startswith("k_sdf_fuse_tiled<false")]
    for mode in ("fast", "fast_tracked"):
        d["full_" + mode] = entry(full, FUSE[mode])
        d["room_" + mode] = entry(room, FUSE[mode], launches=2)
        d["raycast_full_" + mode] = entry(full, RAY[mode])
        d["raycast_room_" + mode] = entry(room, RAY[mode])
    if exact_full:
        d["full_exact"] = entry(full, exact_full[0])
    if exact_room:
        d["room_exact"] = entry(room, exact_room[0])
    d["raycast_full_exact"], d["raycast_room_exact"] = d["raycast_full_fast"], d["raycast_room_fast"]   # the plain march is one kernel in both modes
    d = {k: v for k, v in d.items() if v is not None}
    # which kernels the passes ran on: bench.py reports a figure only while the library it loads is built from the same sources
    # (the summaries and this file are made on the GPU box, from the library that was profiled)
    from kangaroo_amd import _lib
    L = _lib.load()
    d["_kfx_version"] = int(L.kfx_version())
    d["_kernel_source_id"] = {fam: L.kfx_kernel_source_id(fam.encode()).decode() for fam in ("fuse", "raycast")}
    json.dump(d, open(dst, "w"), indent=1)
    for k, v in d.items():
        if isinstance(v, dict) and "traffic_bytes" in v:
            print(k, v["traffic_bytes"])


if __name__ == "__main__":
    main()
