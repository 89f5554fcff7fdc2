// det_math.hpp -- bit-reproducible float64 helpers for host and gfx950 device code.
//
// Only IEEE add/sub/mul (no libm, no contraction: build with -ffp-contract=off) so the
// same inputs give the same bits on x86 and on CDNA4.  Cody-Waite three-constant reduction
// by pi/2 (each constant carries 33 significant bits, so n*Pk is exact for |n| < 2^20)
// followed by the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4].
#pragma once

#if defined(__HIPCC__)
#define ZENV_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define ZENV_HD inline
#endif

namespace zenvk {

ZENV_HD void det_sincos_inl(double x, double &sin_out, double &cos_out)
{
    // nearest multiple of pi/2 via the 1.5*2^52 trick (round-half-even, no rint())
    const double kShift = 6755399441055744.0;
    double fn = (x * 6.36619772367581382433e-01 + kShift) - kShift;
    double r = x - fn * 1.57079632673412561417e+00;
    r = r - fn * 6.07710050630396597660e-11;
    r = r - fn * 2.02226624871116645580e-21;
    r = r - fn * 8.47842766036889956997e-32;
    const long long quadrant = (long long)fn;

    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = -2.50507602534068634195e-08 + z * ps;
    ps = 2.75573137070700676789e-06 + z * ps;
    ps = -1.98412698298579493134e-04 + z * ps;
    ps = 8.33333333332248946124e-03 + z * ps;
    ps = -1.66666666666666324348e-01 + z * ps;
    const double sn = r + r * (z * ps);

    double pc = -1.13596475577881948265e-11;
    pc = 2.08757232129817482790e-09 + z * pc;
    pc = -2.75573143513906633035e-07 + z * pc;
    pc = 2.48015872894767294178e-05 + z * pc;
    pc = -1.38888888888741095749e-03 + z * pc;
    pc = 4.16666666666666019037e-02 + z * pc;
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cs = w + (((1.0 - w) - hz) + (z * z) * pc);

    const int q = (int)(quadrant & 3);
    const double a = (q & 1) ? cs : sn;   // |sin|
    const double b = (q & 1) ? sn : cs;   // |cos|
    sin_out = (q & 2) ? -a : a;
    cos_out = ((q + 1) & 2) ? -b : b;
}

ZENV_HD double det_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

}  // namespace zenvk
