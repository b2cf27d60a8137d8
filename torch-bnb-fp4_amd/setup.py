"""Packaging of the MI355X build of torch_bnb_fp4 (the role of the reference's setup.py:56-83, which drives nvcc).

    pip install -e torch-bnb-fp4_amd --no-build-isolation        # editable: uses the in-tree .so files
    pip install torch-bnb-fp4_amd --no-build-isolation           # regular: copies package + both .so files

The native artefacts are produced by build.py (hipcc, gfx950 only); this file only makes sure they exist and ships them:
`torch_bnb_fp4/lib/libtorch_bnb_fp4_hip.so` as package data and the pybind module `torch_bnb_fp4_ext.so` as a top-level
module next to the package (its rpath is $ORIGIN/torch_bnb_fp4/lib)."""
import os
import shutil
import sys

from setuptools import setup
from setuptools.command.build_py import build_py
from setuptools.command.develop import develop

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _build_native():
    import build as native  # build.py next to this file

    native.build_all(probe=False)  # the product only: tools/stream_probe.hip is bench.py's measuring stick, not shipped
    return native.EXT_LIB


class BuildPyWithNative(build_py):
    def run(self):
        ext = _build_native()
        super().run()
        os.makedirs(self.build_lib, exist_ok=True)
        shutil.copy2(ext, os.path.join(self.build_lib, os.path.basename(ext)))


class DevelopWithNative(develop):
    def run(self):
        _build_native()
        super().run()


setup(
    name="torch-bnb-fp4-amd",
    version="0.1.0",
    description="MI355X (gfx950) HIP kernels behind the torch_bnb_fp4 operator surface: FP4 dequant, fused GEMV, small-batch GEMM, quantiser",
    packages=["torch_bnb_fp4"],
    package_dir={"torch_bnb_fp4": "torch_bnb_fp4"},
    package_data={"torch_bnb_fp4": ["lib/*.so"]},
    python_requires=">=3.9",
    install_requires=[],  # torch (ROCm build) is expected to be present; it is not fetched
    cmdclass={"build_py": BuildPyWithNative, "develop": DevelopWithNative},
    zip_safe=False,
)
