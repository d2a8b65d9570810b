"""Build the reference's OWN furthest-point-sampling kernel for gfx950 into oracle/_ref/ (TEST INFRASTRUCTURE).

    python -m oracle.build_ref            # -> oracle/_ref/libref_fps.so   (needs /root/reference; git-ignored output)

The reference's only native code on a SURVEY 8 row is PointPatchEmbed's sampler,
PointCloud/openpoints/cpp/pointnet2_batch/src/sampling_gpu.cu (furthest_point_sampling_kernel :101-210 and its launcher
:212-258).  It is plain CUDA C that hipcc accepts as-is: the file is compiled WHERE IT LIES under /root/reference -- no copy
of it enters this repository -- with three command-line adjustments and nothing else:
  * -x hip: CUDA C source compiled as HIP;
  * -include oracle/ref_cuda_names.h: the four CUDA runtime names the file uses mapped to their HIP twins;
  * -D_SAMPLING_GPU_H: the include guard of the file's own header, which only declares the torch-tensor wrappers of
    sampling.cpp (not compiled here) and would pull in the torch headers.
tests/test_gpu_pointcloud_ref.py loads the library with ctypes (the launcher's mangled C++ name, default stream) and compares
me_fps' indices with what THIS kernel returns on the same clouds.  Only tests may load it; the GPU box gets the prebuilt file
with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored) and never needs /root/reference.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("METAENC_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF_ROOT, "PointCloud", "openpoints", "cpp", "pointnet2_batch", "src", "sampling_gpu.cu")
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libref_fps.so")
# the launcher furthest_point_sampling_kernel_launcher(int b, int n, int m, const float*, float*, int*) (sampling_gpu.cu:212)
LAUNCHER = "_Z39furthest_point_sampling_kernel_launcheriiiPKfPfPi"


def available() -> bool:
    return os.path.isfile(OUT)


def build(verbose: bool = True) -> str | None:
    if not os.path.isfile(SRC):
        return OUT if available() else None          # (the GPU box: prebuilt file or nothing)
    hipcc = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OUT_DIR, exist_ok=True)
    if available() and os.path.getmtime(OUT) > max(os.path.getmtime(SRC), os.path.getmtime(__file__)):
        return OUT
    cmd = [hipcc, "-O3", "-x", "hip", "--offload-arch=gfx950", "-fPIC", "-shared", "-D_SAMPLING_GPU_H",
           "-include", os.path.join(HERE, "ref_cuda_names.h"), SRC, "-o", OUT]
    if verbose:
        print("[oracle/_ref]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build())
    sys.exit(0 if available() else 1)
