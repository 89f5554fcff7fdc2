"""Instruction mix of the step kernels (CPU only: hipcc -S).  usage: python scripts/isa_stats.py [asm-out] [filter]
Per kernel: instruction count by class (VALU / SALU / LDS / VMEM / SMEM), the spill signature (v_readlane /
v_writelane / scratch), constant re-materialisations (s_mov_b32 of literals), registers."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/zenv_kernels.s"
flt = sys.argv[2] if len(sys.argv) > 2 else "k_rollout_lane|k_step_lane"
src = os.path.join(ROOT, "combinatorial-rl-tasks_amd", "csrc", "kernels.hip")
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                "-Wno-pass-failed", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-o", out, src]
               + os.environ.get("ZENV_EXTRA_FLAGS", "").split(), check=True)
s = open(out).read()
meta = {m.group(1): m.group(2) for m in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", s, re.S)}
print(f"{'kernel':34s} {'insts':>6s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'VMEM':>5s} {'SMEM':>5s} {'rdlane':>6s} {'wrlane':>6s} {'s_mov':>6s} {'scratch':>7s} {'vgpr':>5s} {'sgpr':>5s}")
for m in re.finditer(r"^(_ZN5zenvk\S*?):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not re.search(flt, name):
        continue
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and l.strip() and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    cls = collections.Counter()
    for i, n in c.items():
        k = ("VALU" if i.startswith("v_") else "LDS" if i.startswith("ds_") else "SMEM" if i.startswith("s_load") or i.startswith("s_buffer")
             else "SALU" if i.startswith("s_") else "VMEM")
        cls[k] += n
    t = re.search(r"k_\w+?I(?:Li(\d+)E)(?:Li(\d+)E)?", name)
    short = re.search(r"\d+(k_[a-z_]+)", name).group(1) + (f"<{t.group(1)},{t.group(2)}>" if t else "")
    md = meta.get(name, "")
    vg = re.search(r"next_free_vgpr (\d+)", md); sg = re.search(r"next_free_sgpr (\d+)", md)
    print(f"{short:34s} {len(ins):6d} {cls['VALU']:6d} {cls['SALU']:6d} {cls['LDS']:5d} {cls['VMEM']:5d} {cls['SMEM']:5d} "
          f"{c['v_readlane_b32']:6d} {c['v_writelane_b32']:6d} {c['s_mov_b32']:6d} {sum(n for i, n in c.items() if i.startswith('scratch')):7d} "
          f"{vg.group(1) if vg else '?':>5s} {sg.group(1) if sg else '?':>5s}")
