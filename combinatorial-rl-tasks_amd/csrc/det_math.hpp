// det_math.hpp -- bit-reproducible float64 helpers for host and gfx950 device code.
//
// Only IEEE add/sub/mul and explicitly written fma (no libm, no compiler contraction: build with
// -ffp-contract=off) so the same inputs give the same bits on x86 and on CDNA4.  Cody-Waite three-constant reduction
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
    double fn = __builtin_fma(x, 6.36619772367581382433e-01, kShift) - kShift;
    double r = __builtin_fma(-fn, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-fn, 6.07710050630396597660e-11, r);
    r = __builtin_fma(-fn, 2.02226624871116645580e-21, r);
    r = __builtin_fma(-fn, 8.47842766036889956997e-32, r);
    const int quadrant = (int)fn;   // |x| < 2^31 * pi/2 (a 64-bit conversion costs ~8 instructions per call)

    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(r, z * ps, r);

    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    const double cs = w + __builtin_fma(z * z, pc, (1.0 - w) - hz);

    const int q = (int)(quadrant & 3);
    const double a = (q & 1) ? cs : sn;   // |sin|
    const double b = (q & 1) ? sn : cs;   // |cos|
    sin_out = (q & 2) ? -a : a;
    cos_out = ((q + 1) & 2) ? -b : b;
}

ZENV_HD double det_clamp(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

}  // namespace zenvk
