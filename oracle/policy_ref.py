"""TEST INFRASTRUCTURE -- float32 restatement of the reference's actor network (checker only).

PARITY UNPINNED: the reference ships no checkpoints or fixtures for its models and its modules
(``main/src/env_model.py``, ``policy_network.py``, ``flat_model.py``) import ``gym`` at module top, which
is absent here, so they cannot be imported; this file restates them op for op in torch:

* ``ZoneEnvModel.forward``  main/src/env_model.py:70-79   (zone_net_ :57-63, combine_net_ :65)
* ``PolicyNetwork.forward`` main/src/policy_network.py:40-53 (Box branch), ``fc`` :58-62
* ``ACModel.__init__`` / ``init_params``  main/src/flat_model.py:13-19,24-52 (h_dim 185: utils/agent.py:17)

Only tests/ and bench.py may import it.
"""
import numpy as np
import torch

# (also restated: ACModel's distributional critic, flat_model.py:35-41,56-60)
TENSORS = ("zone_w1", "zone_b1", "zone_w2", "zone_b2", "zone_w3", "zone_b3", "comb_w", "comb_b",
           "enc_w", "enc_b", "mu_w", "mu_b", "std_w", "std_b")


def random_tensors(F, h=185, seed=0, bias_scale=0.1, critic=False, distributional=False):
    """Weights drawn like flat_model.init_params (:13-19: rows of N(0,1) normalised to unit norm); the
    biases, zero there, get small random values so that the bias path is exercised."""
    g = torch.Generator().manual_seed(seed)

    def lin(n_out, n_in):
        w = torch.randn(n_out, n_in, generator=g)
        w = w / torch.sqrt(w.pow(2).sum(1, keepdim=True))
        return w, bias_scale * torch.randn(n_out, generator=g)
    t = {}
    t["zone_w1"], t["zone_b1"] = lin(h, 8 + F)
    t["zone_w2"], t["zone_b2"] = lin(h, h)
    t["zone_w3"], t["zone_b3"] = lin(h, h)
    t["comb_w"], t["comb_b"] = lin(h, 8 + h)
    t["enc_w"], t["enc_b"] = lin(h, h)
    t["mu_w"], t["mu_b"] = lin(2, h)
    t["std_w"], t["std_b"] = lin(2, h)
    if critic or distributional:
        t["critic_w1"], t["critic_b1"] = lin(h, h)          # flat_model.py:43-47 / :35-38
        t["critic_w2"], t["critic_b2"] = lin(1, h)          # critic.2, or critic_mu (:40)
    if distributional:
        t["critic_sigma_w"], t["critic_sigma_b"] = lin(1, h)   # critic_sigma (:41)
    return {k: v.numpy().astype(np.float32) for k, v in t.items()}


def forward_fp32(t, obs, zone_obs):
    """The reference computation, float32.  obs [B,8], zone_obs [B,Z,F] -> mu, std [B,2]."""
    t = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in t.items()}
    obs = torch.as_tensor(obs, dtype=torch.float32)
    zo = torch.as_tensor(zone_obs, dtype=torch.float32)
    bs, n_zones = zo.shape[0], zo.shape[1]
    x = torch.cat([obs.view(bs, 1, 8).expand(bs, n_zones, 8), zo], dim=-1)          # env_model.py:75-78
    x = torch.relu(x @ t["zone_w1"].T + t["zone_b1"])
    x = torch.relu(x @ t["zone_w2"].T + t["zone_b2"])
    x = x @ t["zone_w3"].T + t["zone_b3"]
    zone_emb = x.sum(dim=1) / n_zones
    emb = torch.cat([obs, zone_emb], dim=-1) @ t["comb_w"].T + t["comb_b"]          # :79
    a = torch.relu(emb @ t["enc_w"].T + t["enc_b"])                                  # policy_network.py:48
    mu = 2 * (torch.sigmoid(a @ t["mu_w"].T + t["mu_b"]) - 0.5)                      # :49
    std = torch.sigmoid(a @ t["std_w"].T + t["std_b"]) + 1e-3                        # :50
    if "critic_w1" in t:                                                             # flat_model.py:62-64
        hidden = torch.relu(emb @ t["critic_w1"].T + t["critic_b1"])
        v = (hidden @ t["critic_w2"].T + t["critic_b2"]).squeeze(1)
        if "critic_sigma_w" in t:                                                    # :56-60, Softplus(beta=0.3)
            sigma = torch.nn.functional.softplus(hidden @ t["critic_sigma_w"].T + t["critic_sigma_b"], beta=0.3) + 1e-3
            return mu.numpy(), std.numpy(), v.numpy(), sigma.squeeze(1).numpy()
        return mu.numpy(), std.numpy(), v.numpy()
    return mu.numpy(), std.numpy()


def _round16(x, dtype=torch.bfloat16):
    return x.to(dtype).to(torch.float32)


def forward_bf16_emulated(t, obs, zone_obs, dtype=torch.bfloat16):
    """What the MFMA kernels compute, restated in torch: weights, biases and layer inputs rounded to
    bf16 (dtype=torch.float16: the ZENV_MLP_F16 build of the same kernels), products accumulated in float32 (float64 here; the difference is accumulation order only),
    the zone mean taken after the second ReLU (zone_net_.4 is linear, so it commutes with the mean -- and folds into
    combine_net_, which follows it without an activation)."""
    def _bf(x):
        return _round16(x, dtype)

    raw = {k: torch.as_tensor(v, dtype=torch.float32).double() for k, v in t.items()}
    t = {k: _bf(torch.as_tensor(v, dtype=torch.float32)).double() for k, v in t.items()}
    # zone_net_.4 folded into combine_net_ on the host (float64 products, ONE bf16 rounding): pack_images, mlp_policy.hip
    w_fold = _bf((raw["comb_w"][:, 8:] @ raw["zone_w3"]).float()).double()
    b_fold = _bf((raw["comb_w"][:, 8:] @ raw["zone_b3"] + raw["comb_b"]).float()).double()
    obs = _bf(torch.as_tensor(obs, dtype=torch.float32)).double()
    zo = _bf(torch.as_tensor(zone_obs, dtype=torch.float32)).double()
    bs, n_zones = zo.shape[0], zo.shape[1]
    x = torch.cat([obs.view(bs, 1, 8).expand(bs, n_zones, 8), zo], dim=-1)
    x = _bf(torch.relu(x @ t["zone_w1"].T + t["zone_b1"]).float()).double()
    x = _bf(torch.relu(x @ t["zone_w2"].T + t["zone_b2"]).float())           # bf16 operand of the pooling product
    pooled = (x.sum(dim=1) * np.float32(1.0 / n_zones)).float()              # float32 sum, then * (1/Z)
    c = _bf((_bf(pooled).double() @ w_fold.T + obs @ t["comb_w"][:, :8].T + b_fold).float()).double()
    a = _bf(torch.relu(c @ t["enc_w"].T + t["enc_b"]).float()).double()
    mu = 2 * (torch.sigmoid((a @ t["mu_w"].T + t["mu_b"]).float()) - 0.5)
    std = torch.sigmoid((a @ t["std_w"].T + t["std_b"]).float()) + 1e-3
    if "critic_w1" in t:
        v1 = _bf(torch.relu(c @ t["critic_w1"].T + t["critic_b1"]).float()).double()
        v = (v1 @ t["critic_w2"].T + t["critic_b2"]).float().squeeze(1)
        if "critic_sigma_w" in t:
            sg = torch.nn.functional.softplus((v1 @ t["critic_sigma_w"].T + t["critic_sigma_b"]).float(), beta=0.3) + 1e-3
            return mu.numpy(), std.numpy(), v.numpy(), sg.squeeze(1).numpy()
        return mu.numpy(), std.numpy(), v.numpy()
    return mu.numpy(), std.numpy()
