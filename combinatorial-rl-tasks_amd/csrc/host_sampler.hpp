// host_sampler.hpp -- host half of reset(): NumPy-compatible random streams and the
// Safety-Gym layout rejection sampler, restated for the MI355X zone-env library.
//
// Reference call sites (relative to the reference root):
//   main/envs/TTSP_env.py:19-21            RandomState(seed).beta(3, 1.5) per zone
//   main/envs/colour_match_env.py:57-68    RandomState(seed).choice(colours) per zone
//   main/envs/wrappers.py:10-23            default_rng(rng_seed).integers(min, max+1)
//   [not vendored] safety_gym Engine.reset/build_layout/sample_layout/draw_placement
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/zenv.h"

namespace zenvk {

// numpy.random.RandomState(seed) for an integer seed: MT19937 + the legacy distributions.
class LegacyRandomState {
public:
    explicit LegacyRandomState(uint32_t seed);
    uint32_t next_u32();
    double next_double();
    double uniform(double low, double high);
    int64_t randint_below(int64_t n);       // choice(n) == randint(0, n)
    double beta(double a, double b);        // a, b > 1 branch

private:
    void twist();
    double gauss();
    double standard_gamma(double shape);    // shape > 1
    uint32_t mt_[624];
    int idx_;
    bool has_gauss_;
    double gauss_;
};

// numpy.random.default_rng(seed) restricted to what FixedSeedsWrapper uses.
struct Pcg64State {
    uint64_t state_hi, state_lo, inc_hi, inc_lo;
    uint32_t has_u32, u32;
};
Pcg64State pcg64_from_seed(uint64_t seed);                  // SeedSequence(seed) -> PCG64
uint64_t pcg64_next64(Pcg64State &s);
uint32_t pcg64_next32(Pcg64State &s);
int64_t pcg64_integers(Pcg64State &s, int64_t low, int64_t high_exclusive);

struct Layout {
    double robot_x, robot_y, robot_rot;
    double zone_xy[ZENV_MAX_ZONES][2];
    int32_t aux[ZENV_MAX_ZONES];   // tmax (TimedTSP) / colour 0..2 (ColourMatch) / 0
    int32_t restarts;
};

// Returns 0 or ZENV_E_LAYOUT.
int sample_layout(const zenv_config &cfg, int64_t seed, Layout &out);

// A visiting order for the solver-ordered variant (TSP_order_env.py:49-50 calls OR-tools, which is not available
// here): closed tour from the robot, nearest-neighbour construction (what OR-tools' PATH_CHEAPEST_ARC starts
// from) improved by 2-opt to a local optimum.  rank[z] = position of zone z in the route.  Callers with a real
// solver pass their own ranks through zenv_bank_set's aux column.
void route_ranks(double robot_x, double robot_y, const double (*zone_xy)[2], int Z, int32_t *rank);

// deterministic sin/cos shared in spirit with the device code (same algorithm, same bits)
void det_sincos(double x, double &s, double &c);

// Largest double t with sqrt(t) <= r under round-to-nearest: lets the device test
// d2 <= t instead of sqrt(d2) <= r with identical outcomes.
double sqrt_threshold(double r);

}  // namespace zenvk
