"""The C++ side of the drop-in boundary: the roo:: headers (include/kangaroo/*.h) compile with a
plain host compiler against the C ABI, and the C++ drivers behave (host-only checks on CPU, device
checks on the GPU box)."""
import os
import subprocess

import pytest

import kfx_testlib as T

APPS = os.path.join(T.ROOT, "apps")


def _build():
    subprocess.check_call(["make", "-C", APPS], stdout=subprocess.DEVNULL)


def test_roo_headers_compile_and_host_checks_pass():
    _build()
    out = subprocess.run([os.path.join(APPS, "roo_api_test")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "passed" in out.stdout


def test_reference_signatures_are_declared_verbatim():
    """The operator wrappers keep the reference's call signatures."""
    inc = os.path.join(T.ROOT, "include", "kangaroo")
    fuse = open(os.path.join(inc, "cu_sdffusion.h")).read()
    assert ("void SdfFuse(BoundedVolume<SDF_t> vol, Image<float> depth, Image<float4> norm, Mat<float,3,4> T_cw, "
            "ImageIntrinsics K, float trunc_dist, float maxw, float mincostheta )") in fuse
    assert "void SdfReset(BoundedVolume<SDF_t> vol, float trunc_dist)" in fuse
    assert "void SdfSphere(BoundedVolume<SDF_t> vol, float3 center, float r)" in fuse
    ray = open(os.path.join(inc, "cu_raycast.h")).read()
    assert ("void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_t> vol, "
            "const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, float trunc_dist, bool subpix = true)") in ray
    bil = open(os.path.join(inc, "cu_bilateral.h")).read()
    assert "void BilateralFilter(Image<To> dOut, const Image<Ti> dIn, float gs, float gr, uint size, Ti minval);" in bil
    assert "void NormalsFromVbo(Image<float4> dN, const Image<float4> dV)" in open(os.path.join(inc, "cu_normals.h")).read()
    assert ("void DepthToVbo( Image<float4> dVbo, const Image<T> dKinectDepth, ImageIntrinsics K, float scale = 1.0f);"
            in open(os.path.join(inc, "cu_depth_tools.h")).read())


    fuse1 = " ".join(fuse.split())
    assert ("void SdfFuse( BoundedVolume<SDF_t> vol, BoundedVolume<float> colorVol, Image<float> depth, Image<float4> norm, "
            "Mat<float,3,4> T_cw, ImageIntrinsics K, Image<uchar3> img, Mat<float,3,4> T_iw, ImageIntrinsics Kimg, "
            "float trunc_dist, float max_w, float mincostheta )") in fuse1
    assert "void SdfReset(BoundedVolume<float> vol)" in fuse
    assert ("void RaycastSdf(Image<float> depth, Image<float4> norm, Image<float> img, const BoundedVolume<SDF_t> vol, "
            "const BoundedVolume<float> colorVol, const Mat<float,3,4> T_wc, ImageIntrinsics K, float near, float far, "
            "float trunc_dist, bool subpix = true)") in ray
    assert "void RaycastBox(Image<float> depth, const Mat<float,3,4> T_wc, ImageIntrinsics K, const BoundingBox bbox )" in ray
    assert "void RaycastSphere(Image<float> depth, Image<float> img, const Mat<float,3,4> T_wc, ImageIntrinsics K, float3 center, float r)" in ray
    assert "void RaycastPlane(Image<float> depth, Image<float> img, const Mat<float,3,4> T_wc, ImageIntrinsics K, const float3 n_w )" in ray
    assert ("void SdfDistance(Image<float> dist, Image<float> depth, BoundedVolume<SDF_t> vol, const Mat<float,3,4> T_wc, "
            "ImageIntrinsics K, float trunc_distance)") in fuse
    icp = " ".join(open(os.path.join(inc, "cu_model_refinement.h")).read().split())
    assert ("LeastSquaresSystem<float,6> PoseRefinementProjectiveIcpPointPlane( const Image<float4> dPl, const Image<float4> dPr, "
            "const Image<float4> dNr, const Mat<float,3,4> KT_lr, const Mat<float,3,4> T_rl, float c, "
            "Image<unsigned char> dWorkspace, Image<float4> dDebug )") in icp


def test_host_pose_solve_matches_python():
    """apps/pose_solve.h (C++) and kangaroo_amd/tracking.py (numpy) restate the same Eigen / Sophus steps."""
    import numpy as np
    from kangaroo_amd import tracking
    src = r'''
#include <cstdio>
#include "pose_solve.h"
int main() {
    const double A[36] = {4,1,0,0,2,0, 1,5,1,0,0,0, 0,1,6,1,0,1, 0,0,1,7,1,0, 2,0,0,1,8,1, 0,0,1,0,1,9};
    const double b[6] = {1,-2,3,-4,5,-6};
    double x[6];
    posesolve::FullPivLuSolve<6>(A, b, x);
    for (int i = 0; i < 6; ++i) printf("%.17g\n", x[i]);
    const double t[6] = {0.1,-0.2,0.3,0.2,-0.1,0.15};
    posesolve::SE3d T = posesolve::Exp(t);
    for (int i = 0; i < 3; ++i) printf("%.17g %.17g %.17g %.17g\n", T.R[i][0], T.R[i][1], T.R[i][2], T.t[i]);
    return 0;
}
'''
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", APPS, os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")])
        vals = [float(v) for v in subprocess.check_output([os.path.join(d, "t")], text=True).split()]
    A = np.array([4,1,0,0,2,0, 1,5,1,0,0,0, 0,1,6,1,0,1, 0,0,1,7,1,0, 2,0,0,1,8,1, 0,0,1,0,1,9], float).reshape(6, 6)
    x = tracking.full_piv_lu_solve(A, [1, -2, 3, -4, 5, -6])
    assert np.allclose(vals[:6], x, rtol=1e-13, atol=1e-15)
    Tm = tracking.se3_exp([0.1, -0.2, 0.3, 0.2, -0.1, 0.15])
    assert np.allclose(np.array(vals[6:]).reshape(3, 4), Tm[:3], rtol=1e-13, atol=1e-15)


@pytest.mark.gpu
def test_cpp_api_on_device():
    _build()
    out = subprocess.run([os.path.join(APPS, "roo_api_test")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--fast"], ["--track"], ["--device-icp"], ["--track", "--fused-launches"]])
def test_headless_kinectfusion_app(extra):
    _build()
    out = subprocess.run([os.path.join(APPS, "kinectfusion_headless"), "--res", "128", "--frames", "8"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fps" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--track"], ["--device-icp", "--fused-launches"], ["--track", "--summary"]])
def test_headless_app_recovers_after_a_frame_without_depth(extra):
    """The application's recovery path (main.cpp:223-242) in the C++ loop: --drop-frame 4 delivers one frame without any valid
    depth -> that frame is lost (rmse NaN, nothing fused), the next one resets the model (T_wl = identity, SdfReset(vol, NaN)),
    fuses itself and is tracked from there; the app exits 0 only with exactly one lost frame, one reset and the orbit followed
    (in the frame of the known poses) within 2 cm."""
    _build()
    out = subprocess.run([os.path.join(APPS, "kinectfusion_headless"), "--res", "128", "--frames", "10", "--drop-frame", "4"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "1 frames lost, 1 resets" in out.stdout, out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--track", "--fused-launches"]])
def test_headless_app_with_brick_summary_renders_the_same_images(extra):
    """--summary: roo::SdfSummary through the C++ overloads (tracked SdfReset / SdfFuse on the ROI views of the application,
    RaycastSdf / RaycastSdfLevels through the class tables).  Exact numerics: the last rendering has the bit pattern of the
    run without it -- with known poses and with the ICP loop (whose poses then agree too)."""
    import re
    _build()
    env = dict(os.environ, KFX_RAYCAST_SUMMARY="1")   # always through the tables, whatever share of the volume they cover
    sums = []
    for flag in ([], ["--summary"]):
        out = subprocess.run([os.path.join(APPS, "kinectfusion_headless"), "--res", "128", "--frames", "8"] + extra + flag,
                             capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        m = re.search(r"last raycast hits (\d+)/\d+.*depth checksum ([0-9a-f]{16})", out.stdout)
        assert m and int(m.group(1)) > 1000, out.stdout
        sums.append(m.group(2))
    assert sums[0] == sums[1], sums


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--device-icp", "--fused-launches"]])
def test_headless_app_summary_auto_decides_and_renders_the_same_images(extra):
    """--summary-auto: the application times whole frames in three blocks of eight (tracked pair of kernels, plain pair, tracked
    pair again after roo::SdfSummary::Rebuild), keeps the tables only if both tracked blocks win by 5 % and says which it kept.
    Exact numerics: whichever it keeps, the last rendering has the bit pattern of the run without the summary."""
    import re
    _build()
    sums = []
    for flag in ([], ["--summary-auto"]):
        out = subprocess.run([os.path.join(APPS, "kinectfusion_headless"), "--res", "128", "--frames", "30"] + extra + flag,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        m = re.search(r"last raycast hits (\d+)/\d+.*depth checksum ([0-9a-f]{16})", out.stdout)
        assert m and int(m.group(1)) > 1000, out.stdout
        sums.append(m.group(2))
        if flag:
            d = re.search(r"--summary-auto: frame with the tables ([0-9.]+) / ([0-9.]+) ms, plain kernels ([0-9.]+) ms -> (table|plain) march", out.stdout)
            assert d and float(d.group(1)) > 0 and float(d.group(2)) > 0 and float(d.group(3)) > 0, out.stdout
            assert ("(brick summary)" in out.stdout) == (d.group(4) == "table"), out.stdout
    assert sums[0] == sums[1], sums
