"""In-tree build of libzenv_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libzenv_hip.so")
SOURCES = ["kernels.hip", "mlp_policy.hip", "zenv_api.cpp", "host_sampler.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-pass-failed",
         # MFMA results land in VGPRs where they are converted to the next layer's bf16 operand anyway:
         # without this every accumulator element costs a v_accvgpr_read first (mlp_policy.hip; -20 % VALU)
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _deps():
    out = [os.path.join(_HERE, "..", "include", "zenv.h")]
    for f in os.listdir(CSRC):
        out.append(os.path.join(CSRC, f))
    return out


def build_library(force=False, verbose=False):
    """Compile every HIP/C++ source into lib/libzenv_hip.so; returns its path."""
    if not force and os.path.exists(LIB_PATH):
        if os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(p) for p in _deps()):
            return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    extra = os.environ.get("ZENV_EXTRA_FLAGS", "").split()      # experiments only (e.g. -DZENV_STORE_AUX=16)
    cmd = [_hipcc()] + FLAGS + extra + ["-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
