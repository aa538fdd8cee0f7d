"""Builds doodle_amd/_hostbind*.so in-tree:  python doodle_amd/setup_hostbind.py build_ext --inplace
(invoked by doodle_amd.build.build_hostbind).  Host C++ only — it links libhelio.so."""
import os

from setuptools import setup
from torch.utils.cpp_extension import BuildExtension, CppExtension

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
setup(
    name="doodle_amd_hostbind",
    ext_modules=[CppExtension(
        "doodle_amd._hostbind", [os.path.join("doodle_amd", "csrc", "hostbind.cpp")],
        include_dirs=[os.path.join(ROOT, "include"), "/opt/rocm/include"],
        define_macros=[("__HIP_PLATFORM_AMD__", "1"), ("USE_ROCM", "1")],
        library_dirs=[HERE], libraries=["helio"],
        extra_compile_args=["-O2", "-g0"],
        extra_link_args=["-Wl,-rpath,$ORIGIN"],
    )],
    cmdclass={"build_ext": BuildExtension.with_options(use_ninja=False)},
)
