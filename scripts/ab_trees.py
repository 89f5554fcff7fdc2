"""Diagnostic: same-box A/B of the per-step kernel between two CHECKOUTS of this repository (each with its own package
and built library -- for changes that touch the C ABI, where one Python layer cannot load both libraries).
usage: python scripts/ab_trees.py <treeA> <treeB> [reps]     (e.g. .ab/r03 .  -- a `git worktree` of the last round)
Runs scripts/k1_ab.py's child of each tree alternately."""
import os
import subprocess
import sys

trees = [os.path.abspath(t) for t in sys.argv[1:3]]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
for rep in range(reps):
    for t in trees:
        lib = os.path.join(t, "combinatorial-rl-tasks_amd", "lib", "libzenv_hip.so")
        print(f"--- {t}", flush=True)
        subprocess.run([sys.executable, os.path.join(t, "scripts", "k1_ab.py"), "--child", lib], check=True, cwd=t)
