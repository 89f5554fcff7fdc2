"""In-tree build of libzenv_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libzenv_hip.so")
SOURCES = ["kernels.hip", "mlp_policy.hip", "mlp_policy_f16.hip", "mlp_f32.hip", "zenv_api.cpp", "host_sampler.cpp"]
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


FLAGS_STAMP = os.path.join(LIB_DIR, ".build_flags")


def extra_flags():
    """ZENV_EXTRA_FLAGS: diagnostic variants only (e.g. -DZENV_STORE_AUX=16, -DZENV_EXP=1).  The string is compiled into
    the library (zenv_build_flags()) and stamped beside it, so a variant never passes for the shipped build."""
    return " ".join(os.environ.get("ZENV_EXTRA_FLAGS", "").split())


def _stamped_flags():
    try:
        with open(FLAGS_STAMP) as f:
            return f.read()
    except OSError:
        return None


def _up_to_date():
    """Newer than every source AND compiled with the flag set asked for now (a diagnostic .so left behind by an
    experiment is rebuilt by the next plain build, and the other way round)."""
    return (os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(p) for p in _deps())
            and _stamped_flags() == extra_flags())


def under_profiler():
    """True inside a rocprofv3 / rocprof run: the profiler's preloaded tool library initialises the GPU in every
    process it is injected into, compiler children included, and hipcc's sh -> clang exec hops are then execs
    from a GPU-initialised process -- not allowed on the GPU pool.  Build BEFORE starting the profiler."""
    pre = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in pre or "rocprofiler" in pre:
        return True
    return any(k.startswith(("ROCPROF", "ROCPROFILER_", "ROCP_TOOL", "ROCP_")) for k in os.environ)


def build_library(force=False, verbose=False):
    """Compile every HIP/C++ source into lib/libzenv_hip.so; returns its path.  A no-op when the library is newer
    than every source.  Concurrent callers (the ranks of a torchrun job) are serialised by a file lock, and the
    library appears atomically (compiled beside, then renamed)."""
    if not force and _up_to_date():
        return LIB_PATH
    if under_profiler():
        raise RuntimeError(
            f"{LIB_PATH} is missing or older than csrc/ and this process runs under a ROCm profiler: build first "
            "(`python -c 'import __graft_entry__ as g; g.build()'`), then start rocprofv3")
    import fcntl
    os.makedirs(LIB_DIR, exist_ok=True)
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():          # another rank built it while we waited
                return LIB_PATH
            extra = extra_flags()
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            cmd = [_hipcc()] + FLAGS + extra.split() + ['-DZENV_BUILD_FLAGS="%s"' % extra.replace('"', "'"),
                                                        "-o", tmp] + [os.path.join(CSRC, s) for s in SOURCES]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.run(cmd, check=True)
                os.replace(tmp, LIB_PATH)
                with open(FLAGS_STAMP, "w") as f:
                    f.write(extra)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
