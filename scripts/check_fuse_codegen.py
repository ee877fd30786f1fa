#!/usr/bin/env python3
"""Codegen guard for fuse.hip (CPU only: hipcc cross-compiles gfx950).

The voxel loops of k_sdf_fuse_tiled must not wait for vector memory inside an observation block.  hipcc's register
allocation has produced builds whose loop head reuses the VGPRs of the previous iteration's volume store for the slice
constants read from LDS; the hazard is resolved with `s_waitcnt vmcnt(0)` at the top of every iteration, which drains the
store (and exposes its full latency) before any arithmetic starts: the bit-exact kernel went from 0.438 to 0.486 ms with
an identical instruction sequence otherwise (round 2, found by diffing the two builds).  This script compiles fuse.hip to
assembly with the Makefile's flags and fails if any block with >= 8 ds_read_b128 (an observation of a voxel pair) waits
for vector memory before it has issued a load of its own (i.e. for the previous iteration's traffic), or touches scratch.  Usage: python scripts/check_fuse_codegen.py [path/to/fuse.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kangaroo_amd", "csrc")
FLAGS = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]


def compile_to_asm(out):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", out,
                                      os.path.join(CSRC, "fuse.hip")], check=True, stderr=subprocess.DEVNULL)


def hot_block_waits(asm_path):
    """[(kernel, block label, instruction index, text)] of vmcnt waits inside observation blocks of k_sdf_fuse_tiled."""
    src = open(asm_path).read().split("\n")
    found, kernels = [], 0
    i = 0
    while i < len(src):
        m = re.match(r"^(_ZN3kfx16k_sdf_fuse_tiled\w+):", src[i])
        if not m:
            i += 1
            continue
        kernels += 1
        name, label, block = m.group(1), "entry", []
        i += 1

        def flush():
            if sum(1 for t in block if t.startswith("ds_read_b128")) >= 8:
                loaded = False   # a wait that follows a load of the same block waits for that load: fine
                for k, t in enumerate(block):
                    if t.startswith("global_load") or t.startswith("buffer_load"):
                        loaded = True
                    if t.startswith("s_waitcnt") and "vmcnt" in t and not loaded:
                        found.append((name, label, k, t))
                    if t.startswith("scratch_"):   # a spill inside an observation block: a memory round trip per voxel pair
                        found.append((name, label, k, t))
        while i < len(src) and not src[i].startswith(".Lfunc_end"):
            mm = re.match(r"^(\.LBB\d+_\d+):", src[i])
            if mm:
                flush()
                label, block = mm.group(1), []
            elif src[i].strip() == ";;#ASMSTART":   # hand-written (the keep-loads of the tracked kernels and their wait): not hipcc's doing
                while i < len(src) and src[i].strip() != ";;#ASMEND":
                    i += 1
            elif src[i].startswith("\t") and not src[i].startswith("\t.") and not src[i].startswith("\t;"):
                block.append(src[i].strip())
            i += 1
        flush()
    return kernels, found


def register_budget(asm_path):
    """[(kernel, NumVgprs, ScratchSize, budget)] of the fast two-slice instantiations (k_sdf_fuse_tiled<true, 2, ...>), which
    must fit the 64 VGPRs of 8 waves per SIMD (the 1216-texel tile leaves LDS for 8 workgroups per CU), and of the bit-exact
    untracked ones (<false, ..., false>), which must fit the 80 VGPRs of 6 waves -- both without spilling."""
    out, cur = [], None
    vg = None
    for line in open(asm_path):
        # template arguments: <FAST, ZU, CELL, LX, WY, ZC, TRACK, DXT, NW>: fast two-slice kernels, and bit-exact ones with TRACK = false
        m = re.match(r"^(_ZN3kfx16k_sdf_fuse_tiledIL(?:b1ELi2\w+|b0ELi\d\w+Lb0ELb[01]ELi\d+EEEv\w+)):", line)
        if m:
            cur, vg = m.group(1), None
        elif cur and line.startswith("; NumVgprs:"):
            vg = int(line.split(":")[1])
        elif cur and line.startswith("; ScratchSize:"):
            out.append((cur, vg, int(line.split(":")[1]), 64 if "ILb1E" in cur else 80))
            cur = None
    return out


def keep_load_hazards(asm_path):
    """The tracked fast kernels request the cells of the cached planes with hand-written `global_load_dwordx4 v[a:b], ..., off`
    blocks and wait for them in a later `s_waitcnt vmcnt(0)` block (load_cells, fuse.hip).  hipcc does not know the loads are
    in flight, so nothing it emits between a request and the next such wait may touch the destination registers.  Returns
    (number of requests seen, [(kernel, line number, text)] of instructions that do)."""
    src = open(asm_path).read().split("\n")
    bad, seen = [], 0
    kernel, pending, in_asm = None, {}, False
    for n, line in enumerate(src):
        m = re.match(r"^(_ZN3kfx16k_sdf_fuse_tiled\w+):", line)
        if m:
            kernel, pending = m.group(1), {}
            continue
        if kernel is None:
            continue
        if line.startswith(".Lfunc_end"):
            kernel = None
            continue
        t = line.strip()
        if t == ";;#ASMSTART":
            in_asm = True
            continue
        if t == ";;#ASMEND":
            in_asm = False
            continue
        if not line.startswith("\t") or t.startswith(".") or t.startswith(";"):
            continue
        if in_asm:
            m = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v\[\d+:\d+\], off$", t)
            if m:
                seen += 1
                pending[(int(m.group(1)), int(m.group(2)))] = n
            elif t.startswith("s_waitcnt") and "vmcnt(0)" in t:
                pending = {}
            continue
        if pending:
            regs = set()
            for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t):
                regs.update(range(int(a), int(b) + 1))
            regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t))
            for (a, b) in pending:
                if regs & set(range(a, b + 1)):
                    bad.append((kernel, n + 1, t))
    return seen, bad


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
        kernels, found = hot_block_waits(path)
        budget = register_budget(path)
        keep_seen, keep_bad = keep_load_hazards(path)
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "fuse.s")
            compile_to_asm(path)
            kernels, found = hot_block_waits(path)
            budget = register_budget(path)
            keep_seen, keep_bad = keep_load_hazards(path)
    print("%d k_sdf_fuse_tiled instantiations, %d vector-memory waits inside observation blocks" % (kernels, len(found)))
    for f in found:
        print("  %s %s [%d] %s" % f)
    # ScratchSize: none for the untracked kernels; at most 32 bytes per lane for the fast TRACK instantiations, none of it touched
    # inside an observation block (checked above).  Round 6: the tracked fast kernels keep their uniforms in scalar registers (the
    # bookkeeping of the brick summary takes the VGPRs that would park them), ~40 of which already live in VGPR lanes; with the
    # rectangle from the brick's corners three to eight lane-derived values are parked in scratch around the global-gather fallback
    # loops.  The tiled voxel loops -- where the time goes -- are free of it (S_full through the tracked pair: 0.3471 -> 0.3466 ms,
    # profiles/r06_corner_ab).
    def scratch_limit(name):
        return 32 if re.search(r"ILb1ELi2E\w+Lb1ELb0ELi4EEEv", name) else 0
    over = [b for b in budget if b[1] is None or b[1] > b[3] or b[2] > scratch_limit(b[0])]
    print("%d budgeted instantiations (fast two-slice: 64 VGPRs, bit-exact untracked: 80), %d over the VGPR / scratch budget" % (len(budget), len(over)))
    for b in budget:
        if b in over or b[2]:
            print("  %s NumVgprs %s ScratchSize %s (budget %d)%s" % (b + ("" if b in over else " -- outside the observation blocks",)))
    print("%d hand-written cell requests in the tracked kernels, %d instructions touching their registers before the wait" % (keep_seen, len(keep_bad)))
    for b in keep_bad[:20]:
        print("  %s line %d: %s" % b)
    return 1 if (found or kernels == 0 or over or not budget or keep_seen == 0 or keep_bad) else 0


if __name__ == "__main__":
    sys.exit(main())
