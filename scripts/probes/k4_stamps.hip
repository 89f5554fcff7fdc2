// Diagnostic: where one wave of K4 (k_mlp_zone) spends a row tile.  Builds the kernel file itself with
// -DMLP_STAMP=<wave>, runs it on random weights / rows and prints the median cycles between the stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DMLP_STAMP=0 \
//         -I include scripts/probes/k4_stamps.hip combinatorial-rl-tasks_amd/csrc/mlp_f32.hip combinatorial-rl-tasks_amd/csrc/mlp_policy_f16.hip -o gpurun_out/k4_stamps && gpurun_out/k4_stamps
#include "../../combinatorial-rl-tasks_amd/csrc/mlp_policy.hip"

#include <algorithm>
#include <cstdio>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20;      // timed forwards (power management settles over ~100 ms: use >= 1000 for A/B)
    using namespace zenvk;
    const int N = 65536, Z = 25, F = 6, h = 185;
    std::mt19937 g(1);
    std::normal_distribution<float> nd(0.f, 0.1f);
    auto vec = [&](size_t n) { std::vector<float> v(n); for (auto &x : v) x = nd(g); return v; };
    auto w1 = vec((size_t)h * (8 + F)), b1 = vec(h), w2 = vec((size_t)h * h), b2 = vec(h), w3 = vec((size_t)h * h), b3 = vec(h);
    auto wc = vec((size_t)h * (8 + h)), bc = vec(h), we = vec((size_t)h * h), be = vec(h), wm = vec(2 * h), bm = vec(2), ws = vec(2 * h), bs = vec(2);
    zenv_mlp_weights w{};
    w.h_dim = h;
    w.zone_w1 = w1.data(); w.zone_b1 = b1.data(); w.zone_w2 = w2.data(); w.zone_b2 = b2.data();
    w.zone_w3 = w3.data(); w.zone_b3 = b3.data(); w.comb_w = wc.data(); w.comb_b = bc.data();
    w.enc_w = we.data(); w.enc_b = be.data(); w.mu_w = wm.data(); w.mu_b = bm.data(); w.std_w = ws.data(); w.std_b = bs.data();
    std::vector<uint16_t> img_host;
    size_t offs[8];
    if (pack_images(w, F, img_host, offs) < 0) { printf("pack_images failed\n"); return 1; }
    char *d_img;
    CK(hipMalloc(&d_img, img_host.size() * 2));
    CK(hipMemcpy(d_img, img_host.data(), img_host.size() * 2, hipMemcpyHostToDevice));
    MlpImages img{};
    const void **slots[8] = { &img.w1, &img.w2, &img.w3, &img.wc, &img.wa, &img.wh, &img.wv1, &img.wv2 };
    for (int i = 0; i < 6; ++i) *slots[i] = d_img + offs[i];
    auto obs_h = vec((size_t)N * 8), zo_h = vec((size_t)N * Z * F);
    float *obs, *zo, *mu, *sd;
    void *pooled;
    CK(hipMalloc(&obs, obs_h.size() * 4)); CK(hipMalloc(&zo, zo_h.size() * 4));
    CK(hipMalloc(&mu, N * 8)); CK(hipMalloc(&sd, N * 8)); CK(hipMalloc(&pooled, (size_t)N * kMlpHP * 2));
    CK(hipMemcpy(obs, obs_h.data(), obs_h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(zo, zo_h.data(), zo_h.size() * 4, hipMemcpyHostToDevice));
#ifdef MLP_STAMP
    unsigned long long *d_st;
    CK(hipMalloc(&d_st, 64 * 16 * 8));
    CK(hipMemset(d_st, 0, 64 * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_k4_stamps), &d_st, sizeof(d_st)));
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) CK(launch_mlp_forward(img, N, Z, F, obs, zo, pooled, mu, sd, nullptr, nullptr, no_mlp_action(), nullptr));
    for (int i = 0; i < iters / 2; ++i) CK(launch_mlp_forward(img, N, Z, F, obs, zo, pooled, mu, sd, nullptr, nullptr, no_mlp_action(), nullptr));
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(launch_mlp_forward(img, N, Z, F, obs, zo, pooled, mu, sd, nullptr, nullptr, no_mlp_action(), nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
#ifndef MLP_STAMP
    printf("K4 + K5, no stamps, MLP_EXP=%d: %.1f us per forward\n",
#ifdef MLP_EXP
           (int)(MLP_EXP),
#else
           0,
#endif
           ms / iters * 1e3);
    return 0;
#else
    printf("K4 + K5 with stamps (wave %d): %.1f us per forward\n", (int)(MLP_STAMP), ms / iters * 1e3);
    std::vector<unsigned long long> st(64 * 16);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
#if MLP_KERNEL == 1
    const int n_slots = 7;
#else
    const int n_slots = 4 + NH;
#endif
    const char *names[] = { "tile start -> x0, rows issued, L1 done", "L1 done -> selection matrix", "", "", "", "", "" };
    (void)names;
    printf("slot deltas (cycles, median over tiles 2..49; s_memtime + lgkmcnt(0) at each stamp perturbs):\n");
    for (int sl = 1; sl < n_slots; ++sl) {
        std::vector<long long> d;
        for (int it = 2; it < 50; ++it) d.push_back((long long)(st[it * 16 + sl] - st[it * 16 + sl - 1]));
        std::sort(d.begin(), d.end());
        printf("  stamp %d -> %d: median %6lld   min %6lld   max %6lld\n", sl - 1, sl, d[d.size() / 2], d.front(), d.back());
    }
    std::vector<long long> d;
    for (int it = 3; it < 50; ++it) d.push_back((long long)(st[it * 16] - st[(it - 1) * 16]));
    std::sort(d.begin(), d.end());
    printf("  whole tile (stamp 0 -> next stamp 0): median %lld   min %lld   max %lld\n", d[d.size() / 2], d.front(), d.back());
    if (st[61 * 16] > st[60 * 16])
        printf("  whole tile loop (two stamps per launch): %lld cycles = %.0f per tile, %.3f us, clock %.3f GHz\n",
               (long long)(st[61 * 16] - st[60 * 16]), (double)(st[61 * 16] - st[60 * 16]) / 50.0,
               (double)(st[61 * 16 + 15] - st[60 * 16 + 15]) * 0.01,
               (double)(st[61 * 16] - st[60 * 16]) / ((double)(st[61 * 16 + 15] - st[60 * 16 + 15]) * 10.0));
    printf("  shader clock over tiles 4..44: %.3f GHz\n",
           (double)(st[44 * 16] - st[4 * 16]) / ((double)(st[44 * 16 + 15] - st[4 * 16 + 15]) * 10.0));
    return 0;
#endif
}
