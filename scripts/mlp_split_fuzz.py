"""Random widths / zone counts / batch sizes / weight scales through every network kernel (the split-operand k_mlp_zone_s3,
both halves' kinds, against the torch float32 restatement; the single-product bf16 and float16 builds of k_mlp_zone1 /
k_mlp_head against the restatement with their rounding points) -- a hunt for packing mistakes at odd sizes (h around the
32-feature tile edges, Z = 1, ragged batches), not a timing.  Run on the GPU box: python scripts/mlp_split_fuzz.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ZENV_MLP_F32_MFMA"] = "1"          # the matrix kernel whatever the batch size
import combinatorial_rl_tasks_amd as Z        # noqa: E402
from oracle import policy_ref as P            # noqa: E402  (checker)



def run(cases=40):
    rs = np.random.RandomState(2026)
    worst = {"f16x3": 0.0, "bf16x3": 0.0}
    bad = 0
    for c in range(cases):
        task = int(rs.randint(0, 3))
        zones = int(rs.choice([1, 2, 3, 5, 6, 7, 15, 24, 25, 30]))
        h = int(rs.choice([1, 2, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 95, 96, 97, 127, 128, 129, 160, 185, 190, 191]))
        n = int(rs.choice([1, 31, 32, 33, 63, 64, 65, 127, 200, 257, 1000]))
        scale = float(rs.choice([0.1, 1.0, 3.0]))
        cfg = Z.default_config(task, zones, zones_keepout=0.30 if zones > 15 else 0.45, num_steps=300)
        env = Z.ZoneVecEnv(cfg, n)
        env.build_bank(7 + c, n)
        env.reset()
        env.rollout(int(rs.randint(0, 80)), Z.POLICY_UNIFORM, policy_seed=c)
        dist = bool(rs.randint(0, 2))
        t = P.random_tensors(env.zone_feat, h=h, seed=c, bias_scale=0.1 * scale, critic=True, distributional=dist)
        t = {k: (v * scale if k.endswith("_w") or k.endswith("w1") or k.endswith("w2") else v).astype(np.float32) for k, v in t.items()}
        obs, zo = env.observations()
        ref = P.forward_fp32(t, obs, zo)
        mag = max(1.0, float(np.abs(ref[2]).max()))         # value is unbounded: tolerance relative to its size
        line = "case %2d task %d Z %2d h %3d N %4d scale %.1f dist %d:" % (c, task, zones, h, n, scale, dist)

        def rel_err(prec):
            env.load_mlp(t, precision=prec)
            out = env.mlp_forward(with_value=True)
            err = [float(np.abs(a - b).max()) for a, b in zip(out, ref)]
            return max(err[0], err[1], err[2] / mag, *(err[3:])), all(np.isfinite(a).all() for a in out)
        # the float32 matrix kernel on the same inputs: with big weights (scale 3: activations in the hundreds) float32
        # itself is no longer within 1e-5 of another summation order, and the split modes are held to a multiple of ITS error
        e32, _ = rel_err("f32")
        line += "  f32 %.1e" % e32
        for prec, tol, mult in (("f16x3", 3e-6, 6.0), ("bf16x3", 2e-5, 64.0)):      # (200 cases: worst 4.7 x and 52 x)
            rel, finite = rel_err(prec)
            worst[prec] = max(worst[prec], rel / max(e32, 1e-7))
            ok = finite and rel <= max(tol, mult * e32)
            bad += not ok
            line += "  %s %.1e%s" % (prec, rel, "" if ok else " FAIL")
        # the single-product builds (k_mlp_zone1 / k_mlp_head on bf16 and on float16 operands) against the restatement
        # with their rounding points: a packing mistake at an odd size shows as an error of order one
        import torch
        # (tolerance: the kernel and the restatement round at the same points, but a sum that lands within 1e-7 of a
        # rounding boundary can fall on either side -- one 16-bit ulp in one hidden unit, amplified by the network like the
        # float32 kernel's own noise e32 is: float16's ulp is 2^13 float32 ulps, e32 the sum of a few hundred float32-ulp errors:
        # up to ~1 500 x e32 for float16 (200 cases: worst 760 x), eight times that for bf16)
        for prec, dtype, tol, mult in (("bf16", torch.bfloat16, 4e-3, 12000.0), ("f16", torch.float16, 5e-4, 1500.0)):
            try:
                env.load_mlp(t, precision=prec)
            except Z.ZenvError as ex:              # float16: weights whose bound leaves the range are refused at load
                assert prec == "f16" and ex.code == Z.E_RANGE, ex
                line += "  %s refused" % prec
                continue
            try:
                out = env.mlp_forward(with_value=True)
            except Z.ZenvError as ex:              # ... or an activation of the head at run time
                assert prec == "f16" and ex.code == Z.E_RANGE, ex
                line += "  %s range" % prec
                continue
            emu = P.forward_bf16_emulated(t, obs, zo, dtype=dtype)
            err = [float(np.abs(a - b).max()) for a, b in zip(out, emu)]
            rel = max(err[0], err[1], err[2] / mag, *(err[3:]))
            ok = all(np.isfinite(a).all() for a in out) and rel <= max(tol, mult * e32)
            bad += not ok
            line += "  %s %.1e%s" % (prec, rel, "" if ok else " FAIL %s (|ref| up to %s)" % (
                ["%.1e" % e for e in err], ["%.1e" % float(np.abs(b).max()) for b in emu]))
        print(line, flush=True)
        env.close()
    print("worst error as a multiple of the float32 kernel's:", worst, "| failures:", bad)
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40) else 0)
