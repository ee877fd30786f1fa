"""kangaroo_amd -- MI355X-native KinectFusion volumetric path (SdfFuse / RaycastSdf /
BilateralFilter / DepthToVbo / NormalsFromVbo) behind the C ABI of include/kfx.h.

`kangaroo_amd.roo` mirrors the reference's `roo::` operator interface (same names and
argument meaning) on top of libkfx.so; `kangaroo_amd.scenes` holds the synthetic inputs.
"""
from . import scenes  # noqa: F401

__all__ = ["scenes", "roo"]


def __getattr__(name):
    if name == "roo":
        import importlib
        return importlib.import_module(".roo", __name__)
    raise AttributeError(name)
