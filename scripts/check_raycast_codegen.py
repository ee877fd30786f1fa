#!/usr/bin/env python3
"""Codegen guard for raycast.hip (CPU only: hipcc cross-compiles gfx950) -- round-3 advice.

The march through the class tables requests a sample's cells with a hand-written block of four `global_load_dwordx4`
(RayF32::issue / issue_off32, sampling.h) and waits for them in a LATER hand-written block (`s_waitcnt vmcnt(0)`, finish());
in between the other lanes of the wave consult the tables -- LDS reads, a loop, ~30 live values.  hipcc does not know the
loads are in flight.  The "+v" operands of finish() keep the four destination vectors allocated, but nothing stops the
compiler from COPYING them (a v_mov at a join, a spill to scratch or to an AGPR, a re-materialisation) before the data has
arrived: silently wrong samples, not a fault.  This script compiles raycast.hip to assembly with the Makefile's flags and
fails if, in any kernel, an instruction between such a request block and the next wait block names one of the destination
registers, or if the kernel moves vector registers through scratch or AGPRs at all between the two (every k_raycast_* kernel
is checked; the plain march's single-block loads are complete when their block ends and pass trivially).  It also reports
the kernels' VGPR / SGPR-spill / scratch figures.  Usage: python scripts/check_raycast_codegen.py [path/to/raycast.s]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kangaroo_amd", "csrc")
FLAGS = ["-std=c++17", "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]   # raycast.o in csrc/Makefile


def compile_to_asm(out):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S", "--cuda-device-only", "-o", out,
                                      os.path.join(CSRC, "raycast.hip")], check=True, stderr=subprocess.DEVNULL)


LOAD = re.compile(r"global_load_dwordx4 v\[(\d+):(\d+)\], v(?:\[\d+:\d+\]|\d+), (?:off|s\[\d+:\d+\])")


def regs_named(t):
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", t):
        regs.update(range(int(a), int(b) + 1))
    regs.update(int(a) for a in re.findall(r"\bv(\d+)\b", t))
    return regs


def inflight_hazards(asm_path):
    """(kernels, split requests, [(kernel, line, text, why)]).  Linear scan of each kernel in text order:
      * a hand-written block that ends with loads still in flight ARMS its destination registers; the next hand-written block
        with `s_waitcnt vmcnt(0)` disarms them;
      * while registers are armed, no compiler-emitted instruction may name them -- not as a source (a copy or a spill of data
        that has not arrived) and not as a destination (the returning data would overwrite it) -- and none may move vector
        registers through scratch or AGPRs;
      * the two addressing forms of a request (32-bit offsets from a uniform base / 64-bit addresses: a launch-uniform choice)
        are laid out one after the other, so a second request block that follows the first before any wait is the OTHER
        alternative of the same request: what lies between them is the else-branch's address arithmetic, which runs only when
        the first block did not, and is not held against the first block's registers.
    Text order is not execution order in general; for this loop shape (request, table look-ups of the other lanes, wait,
    blend -- all inside one loop body, the wait block unique) it is."""
    src = open(asm_path).read().split("\n")
    bad, kernels, split_requests = [], 0, 0
    kernel, armed, provisional = None, {}, []
    n = 0
    while n < len(src):
        line = src[n]
        m = re.match(r"^(_ZN3kfx\d+k_raycast\w+):", line)
        if m:
            kernel, armed, provisional, kernels = m.group(1), {}, [], kernels + 1
            n += 1
            continue
        if kernel is None:
            n += 1
            continue
        if line.startswith(".Lfunc_end"):
            bad.extend(provisional)
            if armed:
                bad.append((kernel, n + 1, "", "requests still armed at the end of the kernel (no wait block follows them)"))
            kernel = None
            n += 1
            continue
        t = line.strip()
        if t == ";;#ASMSTART":
            loads, waited = {}, False
            n += 1
            while n < len(src) and src[n].strip() != ";;#ASMEND":
                u = src[n].strip()
                m2 = LOAD.match(u)
                if m2:
                    loads[(int(m2.group(1)), int(m2.group(2)))] = n + 1
                elif u.startswith("s_waitcnt") and "vmcnt(0)" in u:
                    loads, waited = {}, True
                n += 1
            if waited:   # everything requested so far has arrived
                bad.extend(provisional)
                armed, provisional = {}, []
            if loads:
                split_requests += len(loads)
                if armed:   # the other addressing form of the same request: drop what was raised in the else-branch before it
                    provisional = []
                armed = dict(loads)
        elif line.startswith("\t") and not t.startswith(".") and not t.startswith(";") and armed:
            named = regs_named(t)
            for (a, c) in armed:
                if named & set(range(a, c + 1)):
                    provisional.append((kernel, n + 1, t, "names v[%d:%d] while its request is in flight" % (a, c)))
                    break
            if t.startswith("scratch_") or "accvgpr" in t or (t.startswith("buffer_") and "offen" in t):
                provisional.append((kernel, n + 1, t, "vector registers moved through scratch / AGPRs while requests are in flight"))
        n += 1
    return kernels, split_requests, bad


def resources(asm_path):
    out, cur, vals = [], None, {}
    for line in open(asm_path):
        m = re.match(r"^(_ZN3kfx\d+k_raycast\w+):", line)
        if m:
            cur, vals = m.group(1), {}
        elif cur:
            for key in ("NumVgprs", "NumAgprs", "ScratchSize", "sgpr_spill_count", "vgpr_spill_count"):
                mm = re.match(r"^;\s*(?:\.)?%s:\s*(\d+)" % key, line)
                if mm:
                    vals[key] = int(mm.group(1))
            if line.startswith("; ScratchSize:"):
                out.append((cur, dict(vals)))
                cur = None
    return out


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
        kernels, seen, bad = inflight_hazards(path)
        res = resources(path)
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "raycast.s")
            compile_to_asm(path)
            kernels, seen, bad = inflight_hazards(path)
            res = resources(path)
    print("%d k_raycast kernels, %d split cell requests (issue ... finish), %d hazards between request and wait" % (kernels, seen, len(bad)))
    for b in bad[:30]:
        print("  %s line %d: %s  [%s]" % b)
    scratchy = [(k, v) for k, v in res if "classes" in k and v.get("ScratchSize", 0) != 0]
    print("%d class-table kernels, %d with scratch" % (sum(1 for k, _ in res if "classes" in k), len(scratchy)))
    for k, v in res:
        if "classes" in k:
            print("  %s %s" % (k, v))
    return 1 if (bad or kernels == 0 or seen == 0 or scratchy) else 0


if __name__ == "__main__":
    sys.exit(main())
