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


@pytest.mark.gpu
def test_cpp_api_on_device():
    _build()
    out = subprocess.run([os.path.join(APPS, "roo_api_test")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "all checks passed" in out.stdout, out.stdout + out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--fast"]])
def test_headless_kinectfusion_app(extra):
    _build()
    out = subprocess.run([os.path.join(APPS, "kinectfusion_headless"), "--res", "128", "--frames", "8"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "fps" in out.stdout
