#!/usr/bin/env python3
"""stream_parity_512.py -- tests/test_gpu_stream.py's 300-frame comparison (fast numerics through kfx_frame_step against the
exact oracle over the same frames) at the benchmarked size, 512^3 / 640x480, both scenes, tracked; too long for the suite
(the oracle needs ~2 minutes per scene on 128 host threads), run once per round: reports in gpurun_out/stream_parity/.
KFX_STREAM_FRAMES=3000 runs the stream past the saturation of the weights (max_w = 1000: the running average turns into an
exponential one), the regime bench.py's ~5900 untimed frames end in."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kangaroo_amd import roo  # noqa: E402
import test_gpu_stream as S  # noqa: E402

for scene in sys.argv[1:] or ["full", "room"]:
    S.test_gpu_fast_stream_of_300_frames_vs_exact_oracle(roo, scene, 512, 640, 480, True)
    print("stream parity at 512^3:", scene, "ok", flush=True)
