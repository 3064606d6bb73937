"""Builds the PyTorch-ROCm host extension `diff_gaussian_rasterization/_hip_ext*.so` in-tree with g++
(host code only; it links libgsplat_amd.so, which holds every kernel).  Run after the library is built:

    python gaussian-splatting_amd/csrc/torch/build_ext.py
"""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension as ce

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(os.path.dirname(HERE))
NAME = "_hip_ext"
OUT = os.path.join(PKG, "diff_gaussian_rasterization", NAME + sysconfig.get_config_var("EXT_SUFFIX"))
SRC = os.path.join(HERE, "ext.cpp")


def up_to_date():
    deps = [SRC, os.path.join(PKG, "..", "include", "gsplat_amd.h"), __file__]
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps)


def main():
    if up_to_date():
        return
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = ce.include_paths() + [sysconfig.get_paths()["include"], "/opt/rocm/include"]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function", "-Wno-deprecated-declarations",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=" + NAME, "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch.compiled_with_cxx11_abi())]
    cmd += ["-I" + p for p in inc]
    cmd += [SRC, "-o", OUT, "-L" + tlib, "-L" + PKG, "-lgsplat_amd", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
            "-lc10_hip", "-ltorch_python", "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath," + tlib]
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


if __name__ == "__main__":
    main()
    sys.exit(0)
