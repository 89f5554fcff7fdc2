"""Diagnostic: build the library with extra -D flags into gpurun_out/<name>.so (use with ZENV_LIB_PATH)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import combinatorial_rl_tasks_amd.build as B
if B.under_profiler():
    raise SystemExit("this script compiles a variant library: run it without rocprofv3, or build the variant first "
                     "(scripts/build_variant.py) and profile a script that loads it through ZENV_LIB_PATH")
so = os.path.join(ROOT, "gpurun_out", sys.argv[1] + ".so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.run([B._hipcc()] + B.FLAGS + sys.argv[2:] + ["-o", so] + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
print(so)
