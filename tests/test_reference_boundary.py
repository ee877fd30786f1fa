"""Reference-derived checks of the drop-in boundary (build container only: skipped where /root/reference is absent).

(a) Every function prototype the reference declares in include/kangaroo/cu_{sdffusion,raycast,bilateral,normals,
    depth_tools,model_refinement}.h -- read from the reference tree at test time, comments and the export macro removed,
    whitespace removed -- must appear, spelled the same way, in this repo's header of the same name.
(b) The `roo::` statements of the reference application (applications/kinectfusion/main.cpp: container declarations, frame
    pre-amble, reset, raycast per level, ICP, fuse) are pulled by line range into a translation unit generated in a temporary
    directory, with the Sophus / Eigen pose expressions replaced by roo::Mat<float,3,4> placeholders, and compiled
    -fsyntax-only against include/.  Nothing of the reference is stored in the repo: only pass / fail leaves the test.
"""
import os
import re
import subprocess
import tempfile

import pytest

import kfx_testlib as T

REF = "/root/reference"
REF_INC = os.path.join(REF, "include", "kangaroo")
INC = os.path.join(T.ROOT, "include")

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference tree is only present in the build container")

HEADERS = ["cu_sdffusion.h", "cu_raycast.h", "cu_bilateral.h", "cu_normals.h", "cu_depth_tools.h", "cu_model_refinement.h"]


def _strip(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)
    return src.replace("KANGAROO_EXPORT", "")


def _prototypes(path):
    """Declarations `ret name(args);` at namespace scope of a reference header, whitespace removed."""
    src = _strip(open(path).read())
    body = src[src.index("{", src.index("namespace roo")) + 1:src.rindex("}")]
    protos = []
    for decl in body.split(";"):
        decl = decl.strip()
        if "(" not in decl or decl.startswith("struct") or decl.startswith("class") or "{" in decl:
            continue
        protos.append(re.sub(r"\s+", "", decl))
    return protos


# SURVEY.md section 2 (cu_model_refinement.cu row): of that header only the projective point-plane ICP is on the path (section
# 8(f) f-2); the stereo / calibration / ESM pose refinements are OUT OF SCOPE and not declared here
OUT_OF_SCOPE = ("PoseRefinementFromVbo(", "PoseRefinementFromDisparity(", "PoseRefinementFromDisparityESM(", "PoseRefinementFromDepthESM(",
                "CalibrationRgbdFromDepthESM(", "KinectCalibration(", "SumSpeedTest(")


@pytest.mark.parametrize("header", HEADERS)
def test_reference_prototypes_appear_in_the_same_named_header(header):
    protos = _prototypes(os.path.join(REF_INC, header))
    if header == "cu_model_refinement.h":
        protos = [p for p in protos if not any(name in p for name in OUT_OF_SCOPE)]
        assert len(protos) == 1 and "PoseRefinementProjectiveIcpPointPlane(" in protos[0]
    assert protos, "no prototypes found in the reference's " + header
    mine = re.sub(r"\s+", "", _strip(open(os.path.join(INC, "kangaroo", header)).read()))
    # header-inline definitions: `inline` may precede the return type; drop it on both sides of the comparison
    mine = mine.replace("inlinevoid", "void").replace("inlineLeastSquaresSystem", "LeastSquaresSystem")
    missing = [p for p in protos if p not in mine]
    assert not missing, "%s: %d of %d reference prototypes are not declared verbatim: %s" % (header, len(missing), len(protos), missing[:3])


# applications/kinectfusion/main.cpp: the line ranges that hold the application's roo:: statements
# (first line, last line, statements stay at function scope)
MAIN_RANGES = [(104, 119, True), (203, 215, False), (221, 221, True), (225, 240, False), (247, 254, False), (262, 269, False),
               (275, 288, False), (308, 310, False), (347, 355, False)]


def _statements(lines, lo, hi):
    """Statements of lines [lo, hi] (1-based) that mention roo:: (or the volume's voxel size), comments dropped; a statement
    runs from its first line to the terminating ';' (the ICP call spans three lines)."""
    out, cur = [], None
    for ln in lines[lo - 1:hi + 2]:
        code = re.sub(r"//.*", "", ln).strip()
        if not code:
            continue
        if cur is None:
            if "roo::" not in code and "VoxelSizeUnits" not in code:
                continue
            # control-flow heads are not statements: keep only what follows a `)` of an if / for on the same line (none here)
            if re.match(r"(if|for|while)\s*\(", code) or code.startswith("}"):
                continue
            cur = code
        else:
            cur += " " + code
        if cur.rstrip().endswith(";"):
            out.append(cur)
            cur = None
    return out


def _placeholders(stmt):
    # Sophus / Eigen pose expressions -> a roo::Mat<float,3,4> placeholder
    stmt = re.sub(r"\b\w+(?:\.inverse\(\))?\.matrix3x4\(\)", "M34", stmt)              # T_wl.inverse().matrix3x4(), T_wl.matrix3x4()
    stmt = re.sub(r"\((?:[^()]|\([^()]*\))+\)\.matrix3x4\(\)", "M34", stmt)           # (T_cd * T_wl.inverse()).matrix3x4()
    # host frame sources (HAL images) -> plain pointers / sizes
    stmt = re.sub(r"\(([\w\s]+\*)\)\s*images->at\(\d\)->data\(\)", r"(\1)host_ptr", stmt)
    stmt = re.sub(r"images->at\(\d\)->(Width|Height)\(\)", r"host_\1", stmt)
    return stmt


def test_reference_application_statements_compile_against_the_repo_headers():
    lines = open(os.path.join(REF, "applications", "kinectfusion", "main.cpp")).read().split("\n")
    blocks = []
    n_stmt = 0
    for lo, hi, outer in MAIN_RANGES:
        st = [_placeholders(s) for s in _statements(lines, lo, hi)]
        n_stmt += len(st)
        blocks.append((lo, hi, outer, st))
    names = " ".join(s for _, _, _, st in blocks for s in st)
    for fn in ("SdfFuse", "RaycastSdf", "BilateralFilter", "NormalsFromVbo", "DepthToVbo", "SdfReset", "BoxReduceIgnoreInvalid",
               "ElementwiseScaleBias", "PoseRefinementProjectiveIcpPointPlane", "SubBoundingVolume", "Pyramid", "BoundedVolume"):
        assert fn in names, "extraction lost the application's %s statement" % fn
    assert n_stmt >= 30

    tu = ["#include <limits>", "#include <cmath>", "#include <kangaroo/kangaroo.h>", "#include <kangaroo/BoundingBox.h>",
          "#include <kangaroo/Pyramid.h>", "#include <kangaroo/cu_model_refinement.h>", "",
          "void application_statements(int w, int h, int volres, void* host_ptr, int host_Width, int host_Height)", "{",
          "  const int MaxLevels = 4;", "  int its[MaxLevels] = {1, 0, 2, 3};", "  int l = 0;",
          "  bool use_colour = true, showcolor = true;",
          "  float bigs = 1.5f, bigr = 0.1f, knear = 0.4f, kfar = 4.f, trunc_dist_factor = 2.f, max_w = 1000.f, mincostheta = 0.1f, rgb_fl = 535.f, icp_c = 0.1f;",
          "  int biwin = 3;",
          "  roo::ImageIntrinsics K(570.342f, 570.342f, w / 2.0f - 0.5f, h / 2.0f - 0.5f);",
          "  roo::BoundingBox reset_bb(make_float3(-1, -1, 2), make_float3(1, 1, 4));",
          "  roo::Mat<float,3,4> M34, mKT_lp, mT_pl;", "  roo::Mat<roo::ImageKeyframe<uchar3>,10> kfs;", "  unsigned k = 0;",
          "  (void)its; (void)use_colour; (void)showcolor;"]
    for lo, hi, outer, st in blocks:
        if outer:   # the container declarations and trunc_dist stay at function scope
            tu += ["  " + s for s in st]
        else:
            tu += ["  { // main.cpp:%d-%d" % (lo, hi)] + ["    " + s for s in st] + ["  }"]
    tu += ["}", ""]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "kinectfusion_statements.cpp")
        with open(path, "w") as fh:
            fh.write("\n".join(tu))
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-D__HIP_PLATFORM_AMD__", "-I", INC, "-I", "/opt/rocm/include",
                            "-Wno-unused-variable", path], capture_output=True, text=True)
    assert r.returncode == 0, "the application's roo:: statements do not compile against include/:\n" + r.stderr[-4000:]
