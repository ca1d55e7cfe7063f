"""Builds the pybind module `_ext` (csrc/torch_ext.cpp) in-tree with torch.utils.cpp_extension:

    python centerpose_amd/csrc/setup_ext.py build_ext --inplace        (run by __graft_entry__.build())

Host C++ only: the device code lives in libcenterpose_hip.so (hipcc, csrc/Makefile), which the module links against and finds
next to itself at run time (rpath $ORIGIN).  Output: centerpose_amd/_ext.*.so -- the module the reference imports as
`import _ext as _backend` (DCNv2/dcn_v2.py:12) once `centerpose_amd/` is on sys.path or it is registered in sys.modules."""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension, ROCM_HOME

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
os.chdir(PKG)          # build_ext --inplace puts _ext*.so next to libcenterpose_hip.so

setup(name="centerpose_amd_ext",
      ext_modules=[CppExtension("_ext", [os.path.join("csrc", "torch_ext.cpp")],
                                include_dirs=[os.path.join(ROCM_HOME or "/opt/rocm", "include")],
                                define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
                                extra_compile_args=["-O2", "-g0", "-std=c++17", "-Wno-deprecated-declarations"],
                                library_dirs=[PKG], libraries=["centerpose_hip", "c10_hip", "torch_hip"],
                                extra_link_args=["-Wl,-rpath,$ORIGIN", "-s"])],
      cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
      script_args=["build_ext", "--inplace", "--build-temp", os.path.join("csrc", "build", "ext")])
