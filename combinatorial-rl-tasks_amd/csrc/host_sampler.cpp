// host_sampler.cpp -- see host_sampler.hpp.
#include "host_sampler.hpp"

#include <algorithm>
#include <cmath>
#include <vector>
#include <cstring>

#include "det_math.hpp"

namespace zenvk {

// ------------------------------------------------------------------ MT19937 / RandomState
LegacyRandomState::LegacyRandomState(uint32_t seed) : idx_(624), has_gauss_(false), gauss_(0.0)
{
    uint32_t s = seed;
    for (uint32_t i = 0; i < 624; ++i) {
        mt_[i] = s;
        s = 1812433253u * (s ^ (s >> 30)) + i + 1u;
    }
}

void LegacyRandomState::twist()
{
    constexpr int N = 624, M = 397;
    auto mix = [](uint32_t hi, uint32_t lo) {
        uint32_t y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        return (y >> 1) ^ ((lo & 1u) ? 0x9908b0dfu : 0u);
    };
    for (int i = 0; i < N; ++i) {
        mt_[i] = mt_[(i + M) % N] ^ mix(mt_[i], mt_[(i + 1) % N]);
    }
    idx_ = 0;
}

uint32_t LegacyRandomState::next_u32()
{
    if (idx_ >= 624) twist();
    uint32_t y = mt_[idx_++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

double LegacyRandomState::next_double()
{
    const uint32_t hi = next_u32() >> 5;   // 27 bits
    const uint32_t lo = next_u32() >> 6;   // 26 bits
    return (static_cast<double>(hi) * 67108864.0 + static_cast<double>(lo)) / 9007199254740992.0;
}

double LegacyRandomState::uniform(double low, double high)
{
    const double scale = high - low;
    return low + scale * next_double();
}

int64_t LegacyRandomState::randint_below(int64_t n)
{
    const uint64_t top = static_cast<uint64_t>(n) - 1u;
    if (top == 0) return 0;
    uint64_t mask = top;
    for (int sh = 1; sh < 64; sh <<= 1) mask |= mask >> sh;
    for (;;) {
        const uint32_t v = next_u32() & static_cast<uint32_t>(mask);
        if (v <= top) return static_cast<int64_t>(v);
    }
}

double LegacyRandomState::gauss()
{
    if (has_gauss_) {
        has_gauss_ = false;
        const double g = gauss_;
        gauss_ = 0.0;
        return g;
    }
    double x1, x2, r2;
    do {
        x1 = 2.0 * next_double() - 1.0;
        x2 = 2.0 * next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
    gauss_ = f * x1;
    has_gauss_ = true;
    return f * x2;
}

double LegacyRandomState::standard_gamma(double shape)
{
    const double b = shape - 1. / 3.;
    const double c = 1. / std::sqrt(9 * b);
    for (;;) {
        double x, v;
        do {
            x = gauss();
            v = 1.0 + c * x;
        } while (v <= 0.0);
        v = v * v * v;
        const double u = next_double();
        if (u < 1.0 - 0.0331 * (x * x) * (x * x)) return b * v;
        if (std::log(u) < 0.5 * x * x + b * (1. - v + std::log(v))) return b * v;
    }
}

double LegacyRandomState::beta(double a, double b)
{
    const double ga = standard_gamma(a);
    const double gb = standard_gamma(b);
    return ga / (ga + gb);
}

// ------------------------------------------------------------------ SeedSequence + PCG64
namespace {
struct U128 {
    uint64_t hi, lo;
};
inline U128 mul128(U128 a, U128 b)
{
    const unsigned __int128 x = (static_cast<unsigned __int128>(a.hi) << 64) | a.lo;
    const unsigned __int128 y = (static_cast<unsigned __int128>(b.hi) << 64) | b.lo;
    const unsigned __int128 p = x * y;
    return { static_cast<uint64_t>(p >> 64), static_cast<uint64_t>(p) };
}
inline U128 add128(U128 a, U128 b)
{
    U128 r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
constexpr U128 kPcgMult = { 2549297995355413924ull, 4865540595714422341ull };

inline void pcg_step(Pcg64State &s)
{
    U128 st = mul128({ s.state_hi, s.state_lo }, kPcgMult);
    st = add128(st, { s.inc_hi, s.inc_lo });
    s.state_hi = st.hi;
    s.state_lo = st.lo;
}

// numpy SeedSequence (pool of 4 uint32 words)
struct SeedSeq {
    uint32_t pool[4];
    explicit SeedSeq(uint64_t seed)
    {
        uint32_t ent[2];
        int n_ent = 1;
        ent[0] = static_cast<uint32_t>(seed);
        ent[1] = static_cast<uint32_t>(seed >> 32);
        if (ent[1] != 0) n_ent = 2;
        uint32_t hc = 0x43b0d7e5u;
        auto hashmix = [&hc](uint32_t v) {
            v ^= hc;
            hc *= 0x931e8875u;
            v *= hc;
            v ^= v >> 16;
            return v;
        };
        auto mix = [](uint32_t x, uint32_t y) {
            uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
            r ^= r >> 16;
            return r;
        };
        for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
        for (int src = 0; src < 4; ++src)
            for (int dst = 0; dst < 4; ++dst)
                if (src != dst) pool[dst] = mix(pool[dst], hashmix(pool[src]));
    }
    void generate(uint32_t *out, int n) const
    {
        uint32_t hc = 0x8b51f9ddu;
        for (int i = 0; i < n; ++i) {
            uint32_t v = pool[i % 4];
            v ^= hc;
            hc *= 0x58f38dedu;
            v *= hc;
            v ^= v >> 16;
            out[i] = v;
        }
    }
};
}  // namespace

Pcg64State pcg64_from_seed(uint64_t seed)
{
    SeedSeq ss(seed);
    uint32_t w[8];
    ss.generate(w, 8);
    uint64_t v[4];
    for (int i = 0; i < 4; ++i) v[i] = static_cast<uint64_t>(w[2 * i]) | (static_cast<uint64_t>(w[2 * i + 1]) << 32);
    // pcg64_set_seed: initstate = (v0 << 64) | v1, initseq = (v2 << 64) | v3
    Pcg64State s{};
    s.state_hi = 0;
    s.state_lo = 0;
    s.inc_hi = (v[2] << 1) | (v[3] >> 63);
    s.inc_lo = (v[3] << 1) | 1u;
    pcg_step(s);
    U128 st = add128({ s.state_hi, s.state_lo }, { v[0], v[1] });
    s.state_hi = st.hi;
    s.state_lo = st.lo;
    pcg_step(s);
    s.has_u32 = 0;
    s.u32 = 0;
    return s;
}

uint64_t pcg64_next64(Pcg64State &s)
{
    pcg_step(s);
    const uint64_t x = s.state_hi ^ s.state_lo;
    const unsigned rot = static_cast<unsigned>(s.state_hi >> 58);
    return (x >> rot) | (x << ((-rot) & 63u));
}

uint32_t pcg64_next32(Pcg64State &s)
{
    if (s.has_u32) {
        s.has_u32 = 0;
        return s.u32;
    }
    const uint64_t n = pcg64_next64(s);
    s.has_u32 = 1;
    s.u32 = static_cast<uint32_t>(n >> 32);
    return static_cast<uint32_t>(n);
}

int64_t pcg64_integers(Pcg64State &s, int64_t low, int64_t high_exclusive)
{
    // Generator.integers, int64, range < 2^32: Lemire's nearly-divisionless on 32-bit draws
    const uint64_t rng = static_cast<uint64_t>(high_exclusive - 1 - low);
    if (rng == 0) return low;
    const uint32_t rng_excl = static_cast<uint32_t>(rng) + 1u;
    uint64_t m = static_cast<uint64_t>(pcg64_next32(s)) * rng_excl;
    uint32_t leftover = static_cast<uint32_t>(m);
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - static_cast<uint32_t>(rng)) % rng_excl;
        while (leftover < threshold) {
            m = static_cast<uint64_t>(pcg64_next32(s)) * rng_excl;
            leftover = static_cast<uint32_t>(m);
        }
    }
    return low + static_cast<int64_t>(m >> 32);
}

// ------------------------------------------------------------------ layout sampler
int sample_layout(const zenv_config &cfg, int64_t seed, Layout &out)
{
    const int Z = cfg.num_zones;
    std::memset(&out, 0, sizeof(out));

    // task randomness comes from RandomState(seed), i.e. before Engine.reset bumps the seed
    if (cfg.task == ZENV_TASK_TIMED_TSP) {
        LegacyRandomState rs(static_cast<uint32_t>(seed));
        for (int z = 0; z < Z; ++z)
            out.aux[z] = static_cast<int32_t>(rs.beta(cfg.beta_a, cfg.beta_b) * cfg.num_steps);
    } else if (cfg.task == ZENV_TASK_COLOUR_MATCH) {
        // the reference retries up to 100 times but re-seeds identically, so one pass suffices
        LegacyRandomState rs(static_cast<uint32_t>(seed));
        for (int z = 0; z < Z; ++z) out.aux[z] = static_cast<int32_t>(rs.randint_below(3));
    }

    LegacyRandomState rs(static_cast<uint32_t>(seed + 1));
    struct Placed {
        double x, y, keepout;
    } placed[ZENV_MAX_ZONES + 1];
    bool accepted = false;
    for (int attempt = 0; attempt < 10000 && !accepted; ++attempt) {
        int n_placed = 0;
        bool failed = false;
        for (int obj = 0; obj <= Z && !failed; ++obj) {
            const double keepout = (obj == 0) ? cfg.robot_keepout : cfg.zones_keepout;
            // draw_placement: the extents shrunk by the keepout -- or, for an object with a fixed location
            // ('robot_locations', 'zones_locations'), placements_dict_from_object's box (x-k, y-k, x+k, y+k) with
            // k = keepout + 1e-9 shrunk the same way: a 2e-9-wide box that still takes two uniform draws
            double xlo = -cfg.extent + keepout, xhi = cfg.extent - keepout, ylo = xlo, yhi = xhi;
            const double *fixed = nullptr;
            if (obj == 0 && cfg.n_robot_locations > 0) fixed = cfg.robot_location;
            if (obj > 0 && obj - 1 < cfg.n_zones_locations) fixed = cfg.zones_locations[obj - 1];
            if (fixed) {
                const double k = keepout + 1e-9;
                xlo = (fixed[0] - k) + keepout; xhi = (fixed[0] + k) - keepout;
                ylo = (fixed[1] - k) + keepout; yhi = (fixed[1] + k) - keepout;
            }
            bool found = false;
            for (int t = 0; t < 100 && !found; ++t) {
                const double x = rs.uniform(xlo, xhi);
                const double y = rs.uniform(ylo, yhi);
                bool clear = true;
                for (int j = 0; j < n_placed; ++j) {
                    const double ddx = x - placed[j].x, ddy = y - placed[j].y;
                    const double d = std::sqrt(ddx * ddx + ddy * ddy);
                    if (d < placed[j].keepout + cfg.placements_margin + keepout) {
                        clear = false;
                        break;
                    }
                }
                if (clear) {
                    placed[n_placed++] = { x, y, keepout };
                    found = true;
                }
            }
            if (!found) failed = true;
        }
        if (failed) {
            out.restarts++;
        } else {
            accepted = true;
        }
    }
    if (!accepted) return ZENV_E_LAYOUT;
    out.robot_x = placed[0].x;
    out.robot_y = placed[0].y;
    for (int z = 0; z < Z; ++z) {
        out.zone_xy[z][0] = placed[z + 1].x;
        out.zone_xy[z][1] = placed[z + 1].y;
    }
    // build_world_config: robot_rot = random_rot() unless the config fixes it
    out.robot_rot = cfg.robot_rot_fixed ? cfg.robot_rot : rs.uniform(0.0, 2 * 3.141592653589793);
    return 0;
}

void det_sincos(double x, double &s, double &c) { det_sincos_inl(x, s, c); }

void route_ranks(double robot_x, double robot_y, const double (*zone_xy)[2], int Z, int32_t *rank)
{
    // What TSP_Solver.get_optim_route (main/src/utils/TSP_Solver.py:24-62) asks OR-tools for, restated without OR-tools:
    // a closed tour over node 0 = the robot (depot) and nodes 1..Z = the zones, arc cost = the callback's 10 x distance
    // as the routing library's int64 sees it (truncated), first solution PATH_CHEAPEST_ARC, then the default local search
    // (greedy descent, first improvement) to a local optimum.  The library's own operator order and tie-breaks are not
    // reproducible without it: this is the same problem, the same first solution and the same kind of local optimum
    // (relocate, exchange, 2-opt, or-opt of chains up to 3), not its bit-exact route -- a caller who has OR-tools passes
    // its ranks instead (zenv_bank_set / route_fn).
    const int n = Z + 1;
    std::vector<double> x(n), y(n);
    x[0] = robot_x; y[0] = robot_y;
    for (int z = 0; z < Z; ++z) { x[z + 1] = zone_xy[z][0]; y[z + 1] = zone_xy[z][1]; }
    std::vector<int64_t> cost((size_t)n * n);
    for (int a = 0; a < n; ++a)
        for (int b = 0; b < n; ++b)
            cost[(size_t)a * n + b] =
                (int64_t)(std::sqrt((x[a] - x[b]) * (x[a] - x[b]) + (y[a] - y[b]) * (y[a] - y[b])) * 10.0);
    auto d = [&](int a, int b) { return cost[(size_t)a * n + b]; };
    // PATH_CHEAPEST_ARC: extend the path from the depot by the cheapest arc to a free node, lowest index on ties
    std::vector<int> tour(1, 0);
    std::vector<char> used(n, 0);
    used[0] = 1;
    for (int step = 1; step < n; ++step) {
        int best = -1;
        for (int c = 1; c < n; ++c)
            if (!used[c] && (best < 0 || d(tour.back(), c) < d(tour.back(), best))) best = c;
        used[best] = 1;
        tour.push_back(best);
    }
    auto at = [&](int i) { return tour[i % n]; };            // closed: position n is the depot again
    auto length = [&]() {
        int64_t s = 0;
        for (int i = 0; i < n; ++i) s += d(at(i), at(i + 1));
        return s;
    };
    bool improved = n > 3;
    while (improved) {
        improved = false;
        // or-opt / relocate: the chain tour[i .. i + len - 1] moves between tour[j] and tour[j + 1], orientation kept
        for (int len = 1; len <= 3 && !improved; ++len)
            for (int i = 1; i + len <= n && !improved; ++i)
                for (int j = 0; j < n && !improved; ++j) {
                    if (j >= i - 1 && j < i + len) continue;               // the chain itself or its own predecessor
                    const int p = at(i - 1), f = at(i), l = at(i + len - 1), q = at(i + len), a = at(j), b = at(j + 1);
                    const int64_t gain = d(p, f) + d(l, q) + d(a, b) - d(p, q) - d(a, f) - d(l, b);
                    if (gain <= 0) continue;
                    std::vector<int> chain(tour.begin() + i, tour.begin() + i + len), rest;
                    for (int k = 0; k < n; ++k)
                        if (k < i || k >= i + len) rest.push_back(tour[k]);
                    const int where = (int)(std::find(rest.begin(), rest.end(), a) - rest.begin()) + 1;
                    rest.insert(rest.begin() + where, chain.begin(), chain.end());
                    tour = rest;
                    improved = true;
                }
        // exchange: two zones swap places
        for (int i = 1; i < n && !improved; ++i)
            for (int j = i + 1; j < n && !improved; ++j) {
                const int64_t before = length();
                std::swap(tour[i], tour[j]);
                if (length() < before) improved = true;
                else std::swap(tour[i], tour[j]);
            }
        // 2-opt: reverse tour[i .. j]
        for (int i = 1; i < n - 1 && !improved; ++i)
            for (int j = i + 1; j < n && !improved; ++j) {
                const int a = at(i - 1), b = at(i), c = at(j), e = at(j + 1);
                if (d(a, c) + d(b, e) < d(a, b) + d(c, e)) {
                    std::reverse(tour.begin() + i, tour.begin() + j + 1);
                    improved = true;
                }
            }
    }
    for (int k = 1; k < n; ++k) rank[tour[k] - 1] = k - 1;
}

double sqrt_threshold(double r)
{
    double t = r * r;
    while (std::sqrt(t) > r) t = std::nextafter(t, 0.0);
    for (;;) {
        const double up = std::nextafter(t, INFINITY);
        if (std::sqrt(up) <= r) t = up;
        else break;
    }
    return t;
}

}  // namespace zenvk
