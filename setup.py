"""Build / install of the MI355X-native rasterizer as a PyTorch-ROCm extension.

    python setup.py build_ext --inplace        # in-tree (what __graft_entry__.build() runs)
    pip install --no-build-isolation -e .      # or a regular install

Two native artefacts, both for gfx950 only:

  frosting_amd/lib/libfrosting_rasterizer.so   every hand-written HIP kernel + the C ABI of
                                               include/frosting_rasterizer.h, compiled by hipcc through
                                               frosting_amd/csrc/Makefile (per-file floating-point
                                               contraction flags are part of the arithmetic contract,
                                               which a single extra_compile_args list cannot express);
  diff_gaussian_rasterization/_C.*.so          the pybind module the reference's Python imports
                                               (DGR/setup.py:21-29 builds the same name): torch tensors
                                               in, the C ABI underneath, torch's current HIP stream.

The reference's counterpart is gaussian_splatting/submodules/diff-gaussian-rasterization/setup.py.
"""
import os
import subprocess

from setuptools import find_packages, setup
from setuptools.command.build_py import build_py
from torch.utils.cpp_extension import BuildExtension, CppExtension

ROOT = os.path.dirname(os.path.abspath(__file__))
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
CSRC = os.path.join("frosting_amd", "csrc")


def make_hip_library():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, CSRC), "-j8", "ARCH=gfx950"])


class BuildHipThenBinding(BuildExtension):
    """hipcc (Makefile) first: the binding links against the library it produces."""

    def run(self):
        make_hip_library()
        super().run()


class BuildPyWithHipLibrary(build_py):
    """`pip install .` runs build_py BEFORE build_ext: package_data's lib/*.so is collected here, so the HIP library
    must exist by now -- otherwise a non-editable install ships _C without the library its rpath points at."""

    def run(self):
        make_hip_library()
        super().run()


binding = CppExtension(
    name="diff_gaussian_rasterization._C",
    sources=[os.path.join(CSRC, "torch_ext", "torch_binding.cpp")],
    include_dirs=[os.path.join(ROOT, "include"), os.path.join(ROCM, "include")],
    define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
    library_dirs=[os.path.join(ROOT, "frosting_amd", "lib"), os.path.join(ROCM, "lib")],
    libraries=["frosting_rasterizer", "c10_hip", "torch_hip", "amdhip64"],
    extra_compile_args=["-O2", "-std=c++17", "-Wno-unused-function"],
    # _C lives in diff_gaussian_rasterization/, the HIP library in frosting_amd/lib/ beside it
    extra_link_args=["-Wl,-rpath,$ORIGIN/../frosting_amd/lib", "-Wl,-rpath," + os.path.join(ROCM, "lib")],
)

setup(
    name="diff_gaussian_rasterization",
    version="0.2.0",
    description="MI355X-native differentiable Gaussian-splat rasterizer (drop-in for diff_gaussian_rasterization)",
    packages=find_packages(include=["diff_gaussian_rasterization", "frosting_amd", "frosting_amd.*"]),
    package_data={"frosting_amd": ["lib/*.so"]},
    ext_modules=[binding],
    cmdclass={"build_ext": BuildHipThenBinding.with_options(use_ninja=False), "build_py": BuildPyWithHipLibrary},
    zip_safe=False,
)
