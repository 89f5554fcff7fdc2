"""CPU emulation (torch): how accurate would a split-bf16 policy forward be?  

Every operand of every layer -- weights and the activations feeding the layer -- is written as a sum of k bf16 terms
(x = x1 + x2 + ..., x1 = bf16(x), x2 = bf16(x - x1), ...), the products of terms are accumulated in float32 as the MFMA
does, and the terms whose combined order exceeds `order` are dropped:
    k = 1              1 product   (the shipped bf16 kernels K4 / K5)
    k = 2, order 3     3 products  hi*hi + hi*lo + lo*hi            ("bf16x3")
    k = 2, order 4     4 products  + lo*lo
    k = 3, order 4     6 products  hh hm mh hl lh mm                 ("bf16x6")
    k = 3, order 6     9 products
Reported: max |mu|, |std|, |value| error against the float32 restatement of the reference's modules
(oracle/policy_ref.py forward_fp32) over observations drawn like the env's, and the MFMA count per zone-row tile that
goes with it (the shipped K4 issues 90 per 32 rows and is matrix-pipe bound at ~110 us per step of 65 536 x 25 rows).
Test/diagnostic infrastructure: imports the checker in oracle/, never shipped."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import policy_ref as P   # noqa: E402


ELEM = torch.bfloat16       # main() also runs the table with torch.float16 (11 significant bits per term, subnormals kept)


def split(x, k):
    terms, r = [], x.float()
    for _ in range(k):
        t = r.to(ELEM).float()
        terms.append(t)
        r = r - t
    return terms


def matmul_split(x, w, k, order):
    """x [.., K] @ w[N, K].T with both operands split into k bf16 terms; products of order i + j + 2 <= order kept;
    each product accumulated in float32 (emulated in float64 and rounded: accumulation order aside, what MFMA gives)."""
    xs, ws = split(x, k), split(w, k)
    acc = torch.zeros(x.shape[:-1] + (w.shape[0],), dtype=torch.float64)
    n = 0
    for i, xt in enumerate(xs):
        for j, wt in enumerate(ws):
            if i + j + 2 <= order:
                acc += xt.double() @ wt.double().T
                n += 1
    return acc.float(), n


def forward_split(t, obs, zone_obs, k, order):
    t = {kk: torch.as_tensor(v, dtype=torch.float32) for kk, v in t.items()}
    obs = torch.as_tensor(obs, dtype=torch.float32)
    zo = torch.as_tensor(zone_obs, dtype=torch.float32)
    bs, nz = zo.shape[0], zo.shape[1]
    ones = lambda x: torch.cat([x, torch.ones(x.shape[:-1] + (1,))], dim=-1)   # noqa: E731  the bias as a weight column

    def lin(x, w, b):
        y, n = matmul_split(ones(x), torch.cat([w, b[:, None]], dim=1), k, order)
        return y, n
    x = torch.cat([obs.view(bs, 1, 8).expand(bs, nz, 8), zo], dim=-1)
    x, n1 = lin(x, t["zone_w1"], t["zone_b1"]); x = torch.relu(x)
    x, n2 = lin(x, t["zone_w2"], t["zone_b2"]); x = torch.relu(x)
    pooled = x.sum(dim=1) * np.float32(1.0 / nz)
    z3, _ = lin(pooled, t["zone_w3"], t["zone_b3"])
    emb, _ = lin(torch.cat([obs, z3], dim=-1), t["comb_w"], t["comb_b"])
    a, _ = lin(emb, t["enc_w"], t["enc_b"]); a = torch.relu(a)
    mu = 2 * (torch.sigmoid(lin(a, t["mu_w"], t["mu_b"])[0]) - 0.5)
    std = torch.sigmoid(lin(a, t["std_w"], t["std_b"])[0]) + 1e-3
    hid = torch.relu(lin(emb, t["critic_w1"], t["critic_b1"])[0])
    v = lin(hid, t["critic_w2"], t["critic_b2"])[0].squeeze(1)
    return mu.numpy(), std.numpy(), v.numpy(), n2


def main():
    rs = np.random.RandomState(0)
    B, Z, F = 512, 25, 6
    obs = np.concatenate([rs.uniform(0, 1, (B, 1)), rs.uniform(-1, 1, (B, 7))], 1).astype(np.float32)
    zo = np.concatenate([rs.uniform(-1, 1, (B, Z, 2)), rs.randint(0, 2, (B, Z, 3)), np.full((B, Z, 1), 0.25)], 2).astype(np.float32)
    print(f"{'variant':28s} {'products':>8s} {'MFMA/tile':>9s} {'max|d mu|':>10s} {'max|d std|':>10s} {'max|d value|':>12s}")
    global ELEM
    for seed in (0, 1):
        t = P.random_tensors(F, h=185, seed=seed, critic=True)
        mu0, std0, v0 = P.forward_fp32(t, obs, zo)
        for elem, ename in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
            ELEM = elem
            for name, k, order in (("%s", 1, 2), ("%sx3: hh + hl + lh", 2, 3), ("%sx4: + ll", 2, 4),
                                   ("%sx6: 3 terms, order 4", 3, 4), ("%sx9: 3 terms, all", 3, 6)):
                mu, std, v, n = forward_split(t, obs, zo, k, order)
                print(f"{name % ename:28s} {n:8d} {90 * n:9d} {np.abs(mu - mu0).max():10.2e} {np.abs(std - std0).max():10.2e} "
                      f"{np.abs(v - v0).max():12.2e}   (weights seed {seed})")


if __name__ == "__main__":
    main()
