"""Experiment: the neural-policy rollout of 65 536 envs as ONE handle vs TWO handles of 32 768 envs on their own
streams, driven by two host threads -- does the latency-bound env step of one half hide behind the other half's
matrix kernel?"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import combinatorial_rl_tasks_amd as Z

def weights(F, h=185, seed=0):
    rs = np.random.RandomState(seed)
    def lin(n_out, n_in):
        w = rs.standard_normal((n_out, n_in)).astype(np.float32)
        return w / np.sqrt((w * w).sum(1, keepdims=True)), (0.1 * rs.standard_normal(n_out)).astype(np.float32)
    t = {}
    for (kw, kb), shape in ((("zone_w1", "zone_b1"), (h, 8 + F)), (("zone_w2", "zone_b2"), (h, h)), (("zone_w3", "zone_b3"), (h, h)),
                            (("comb_w", "comb_b"), (h, 8 + h)), (("enc_w", "enc_b"), (h, h)), (("mu_w", "mu_b"), (2, h)),
                            (("std_w", "std_b"), (2, h))):
        t[kw], t[kb] = lin(*shape)
    return t

def make(n, first):
    cfg = Z.default_config(0, 25, zones_keepout=0.40)
    env = Z.ZoneVecEnv(cfg, n); env.build_bank(1 + first, n, n_threads=16); env.reset(); env.load_mlp(weights(6), precision="bf16")
    env.rollout(300, Z.POLICY_MLP_MEAN)
    return env

T = 300
one = make(65536, 0)
t0 = time.perf_counter(); one.rollout(T, Z.POLICY_MLP_MEAN); t1 = time.perf_counter() - t0
print("one handle, 65536 envs: %.1f us per step" % (t1 / T * 1e6))
one.close()
halves = [make(32768, 0), make(32768, 32768)]
def run(e): e.rollout(T, Z.POLICY_MLP_MEAN)
for rep in range(2):
    th = [threading.Thread(target=run, args=(e,)) for e in halves]
    t0 = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    t2 = time.perf_counter() - t0
    print("two handles x 32768 envs, two streams: %.1f us per step of the 65536" % (t2 / T * 1e6))
