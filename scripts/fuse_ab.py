#!/usr/bin/env python3
"""A/B timing + agreement of SdfFuse kernel variants at 512^3 on both synthetic scenes.

A variant is `mode[:ENV=val[,ENV=val...]]` with mode in {exact, fast}; the environment variables
select experimental kernels inside libkfx (KFX_FUSE_TILED, KFX_FUSE_CAP, KFX_FUSE_ZC, KFX_FUSE_KERNEL,
KFX_FUSE_FAST_VARIANT).  Each variant runs in its own process.  Exact variants are compared by a
checksum of the volume bits (must all be identical); fast variants by L-inf / classification flips
against the default exact kernel, in-process.

Usage: python scripts/fuse_ab.py exact:KFX_FUSE_TILED=0 exact fast:KFX_FUSE_TILED=0 fast"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(mode, N=512, w=640, h=480, reps=20):
    import torch
    from kangaroo_amd import roo, scenes
    out = {}
    fast = mode == "fast"
    for scene in ("full", "room"):
        bmin, bmax, near, far = scenes.SCENES[scene]
        K = scenes.intrinsics(w, h)
        tr = scenes.trunc_dist(bmin, bmax, (N, N, N))
        raw = roo.Image(w, h).MemcpyFromHost(scenes.render_depth(scene, w, h, None, K))
        f, vbo, nrm = roo.Image(w, h), roo.Image(w, h, "f32x4"), roo.Image(w, h, "f32x4")
        roo.BilateralFilter(f, raw, **scenes.BILATERAL)
        roo.DepthToVbo(vbo, f, K)
        roo.NormalsFromVbo(nrm, vbo)
        poses = [scenes.se3_inverse(scenes.orbit_pose(i, 30)) for i in range(4)]

        def run(m):
            roo.set_math_mode(m)
            vol = roo.BoundedVolume(N, N, N, bmin, bmax)
            roo.SdfReset(vol, float("nan"))
            for i in range(3):
                roo.SdfFuse(vol, f, nrm, poses[i], K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            torch.cuda.synchronize()
            return vol

        vol = run(mode)
        if fast:
            ref = run("exact")
            a, b = vol.tensor(), ref.tensor()
            na, nb = torch.isnan(a[..., 0]), torch.isnan(b[..., 0])
            both = ~na & ~nb
            d = (a[..., 0][both] - b[..., 0][both]).abs()
            out[scene] = {"flips": int((na != nb).sum()), "linf": float(d.max()), "n>1e-4": int((d > 1e-4).sum()),
                          "n>1e-6": int((d > 1e-6).sum())}
            del ref, a, b, d
            roo.set_math_mode("fast")
        else:
            t = vol.tensor().contiguous().view(torch.int32).to(torch.int64)
            out[scene] = {"checksum": int(t.sum().item()) ^ int((t * torch.arange(1, 3, device=t.device)).sum().item())}
            del t
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a_, b_ in ev:
            a_.record()
            roo.SdfFuse(vol, f, nrm, poses[3], K, tr, scenes.MAX_W, scenes.MIN_COS_THETA)
            b_.record()
        torch.cuda.synchronize()
        ms = sorted(a_.elapsed_time(b_) for a_, b_ in ev)
        out[scene]["ms_med"] = round(ms[len(ms) // 2], 4)
        out[scene]["ms_min"] = round(ms[0], 4)
        del vol
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        variants = sys.argv[1:] or ["exact:KFX_FUSE_TILED=0", "exact", "fast:KFX_FUSE_TILED=0", "fast"]
        base = None
        for v in variants:
            mode, _, envs = v.partition(":")
            env = dict(os.environ)
            for kv in filter(None, envs.split(",")):
                k, _, val = kv.partition("=")
                env[k] = val
            p = subprocess.run([sys.executable, __file__, "--child", mode], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("%-44s FAILED %s" % (v, p.stderr[-1500:]))
                continue
            r = json.loads(line[0][7:])
            note = ""
            if mode == "exact":
                cs = (r["full"]["checksum"], r["room"]["checksum"])
                if base is None:
                    base = cs
                note = "bit-identical to first exact: %s" % (cs == base)
                txt = "full %.4f/%.4f ms  room %.4f/%.4f ms" % (r["full"]["ms_med"], r["full"]["ms_min"], r["room"]["ms_med"], r["room"]["ms_min"])
            else:
                txt = "full %.4f/%.4f ms  room %.4f/%.4f ms" % (r["full"]["ms_med"], r["full"]["ms_min"], r["room"]["ms_med"], r["room"]["ms_min"])
                note = "vs exact: full flips=%d linf=%.2e n>1e-4=%d n>1e-6=%d | room flips=%d linf=%.2e n>1e-4=%d n>1e-6=%d" % (
                    r["full"]["flips"], r["full"]["linf"], r["full"]["n>1e-4"], r["full"]["n>1e-6"],
                    r["room"]["flips"], r["room"]["linf"], r["room"]["n>1e-4"], r["room"]["n>1e-6"])
            print("%-44s %s  [%s]" % (v, txt, note), flush=True)
