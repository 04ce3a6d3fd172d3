// ca_math.h -- deterministic scalar helpers used by the HIP kernels and by the host-side scenario
// generators of libcaenv.so.  Every function is a fixed sequence of single IEEE operations
// (the library is built with -ffp-contract=off), so host and device agree bit for bit and the
// results do not depend on any libm.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CA_HD __host__ __device__ __forceinline__

namespace ca {

// ---- fp32 2-vectors; operation order is that of the RVO2 library's Vector2 ----
struct V2 {
    float x, y;
};
CA_HD V2 mk(float x, float y) { V2 v; v.x = x; v.y = y; return v; }
CA_HD V2 operator+(V2 a, V2 b) { return mk(a.x + b.x, a.y + b.y); }
CA_HD V2 operator-(V2 a, V2 b) { return mk(a.x - b.x, a.y - b.y); }
CA_HD V2 operator-(V2 a) { return mk(-a.x, -a.y); }
CA_HD V2 operator*(float s, V2 a) { return mk(s * a.x, s * a.y); }
CA_HD V2 operator*(V2 a, float s) { return mk(a.x * s, a.y * s); }
CA_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
CA_HD float det(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
CA_HD float absSq(V2 a) { return a.x * a.x + a.y * a.y; }
CA_HD float sqr(float a) { return a * a; }
CA_HD float vabs(V2 a) { return sqrtf(absSq(a)); }
// a / b, IEEE-correct, for operands AWAY FROM THE ENDS OF THE EXPONENT RANGE.  The compiler's correctly rounded fp32 division is
// eleven instructions on gfx950: v_div_scale x 2 (pre-scale numerator / denominator by 2^+-64 when the quotient would leave the
// normal range), v_rcp, six FMAs of Newton refinement and residual correction (the last one v_div_fmas, which undoes the scaling),
// v_div_fixup (infinities, NaNs, zeros).  For 2^-60 <= |b| <= 2^60 and a = 0 or 2^-60 <= |a| <= 2^60 the scale and fix-up
// instructions are the identity, so the remaining eight ARE that sequence and return its bits -- the correctly rounded quotient,
// what the oracle's host division returns.  Used where the operands of every lane whose result is USED are in range by
// construction (the callers say why); a lane outside the range gets some value or NaN and must not use it.
// tests/test_gpu_parity.py::test_numerics_contract_division_in_range: 2e9 pairs against the host's division.
CA_HD float div_ir(float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e0 = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e0, r0, r0);
    const float q0 = a * r1;
    const float e1 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(e2, r1, q1);
#else
    return a / b;
#endif
}
// sqrt(x), IEEE-correct, for x = 0 or x >= 2^-96: the compiler's correctly rounded fp32 square root (v_sqrt_f32, then the
// choice among the result and its two neighbours by the signs of two FMA residuals) without its pre-scaling of tiny arguments
// and its class fix-up: 9 instructions instead of 16.  A correctly rounded result is unique, so these are the bits of the host's
// sqrtf.  (0, +inf and NaN come out right as well; a negative argument gives a NaN.)
// tests/test_gpu_parity.py::test_numerics_contract_sqrt_in_range.
CA_HD float sqrt_ir(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __uint_as_float(__float_as_uint(y) - 1u), yp = __uint_as_float(__float_as_uint(y) + 1u);
    const float rm = __builtin_fmaf(-ym, y, x), rp = __builtin_fmaf(-yp, y, x);
    y = (rm <= 0.0f) ? ym : y;
    y = (rp > 0.0f) ? yp : y;
    return y;
#else
    return sqrtf(x);
#endif
}
CA_HD V2 vdiv(V2 a, float s) {  // vector / scalar multiplies by the reciprocal
    const float inv = 1.0f / s;
    return mk(a.x * inv, a.y * inv);
}
// ... with the reciprocal through div_ir: for s in [2^-60, 2^60] (or a result that is not used)
CA_HD V2 vdiv_ir(V2 a, float s) {
    const float inv = div_ir(1.0f, s);
    return mk(a.x * inv, a.y * inv);
}
CA_HD V2 normalize_ir(V2 a) { return vdiv_ir(a, sqrt_ir(a.x * a.x + a.y * a.y)); }   // (|a| = 0 -- NaN either way -- or in [2^-48, 2^48])
CA_HD V2 normalize(V2 a) { return vdiv(a, vabs(a)); }
CA_HD float leftOf(V2 a, V2 b, V2 c) { return det(a - c, b - a); }
// (the segment is an obstacle edge: ca_set_obstacles refuses edges of length zero, so the division is in div_ir's range)
CA_HD float distSqPointSegment(V2 a, V2 b, V2 c) {
    const float r = div_ir(dot(c - a, b - a), absSq(b - a));
    if (r < 0.0f) return absSq(c - a);
    if (r > 1.0f) return absSq(c - b);
    return absSq(c - (a + r * (b - a)));
}

// ---- sin/cos in fp64: quadrant reduction + fixed polynomials (|a| <~ 1e5, error <~ 2e-16) ----
CA_HD void sincos64(double a, double* s, double* c) {
    const double kd = floor(a * 6.36619772367581382433e-01 + 0.5);
    const double r = (a - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double sp = r + (z * r) * (-1.66666666666666324348e-01 +
                      z * (8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)))));
    const double cp = 1.0 - (0.5 * z - (z * z) * (4.16666666666666019037e-02 +
                      z * (-1.38888888888741095749e-03 +
                      z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 +
                      z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))))));
    const int q = (int)((long long)kd & 3);
    *s = (q == 0) ? sp : (q == 1) ? cp : (q == 2) ? -sp : -cp;
    *c = (q == 0) ? cp : (q == 1) ? -sp : (q == 2) ? -cp : sp;
}

// ---- e^x in fp64: k = round(x/ln2), r = x - k ln2 in two parts, degree-13 Taylor polynomial in
// Horner form, exact scaling by 2^k (|x| <~ 700, error <~ 2e-16 relative) ----
CA_HD double exp64(double x) {
    const double kd = floor(x * 1.44269504088896338700e+00 + 0.5);
    const double r = (x - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    double pl = 1.0 / 6227020800.0;
    pl = pl * r + 1.0 / 479001600.0;
    pl = pl * r + 1.0 / 39916800.0;
    pl = pl * r + 1.0 / 3628800.0;
    pl = pl * r + 1.0 / 362880.0;
    pl = pl * r + 1.0 / 40320.0;
    pl = pl * r + 1.0 / 5040.0;
    pl = pl * r + 1.0 / 720.0;
    pl = pl * r + 1.0 / 120.0;
    pl = pl * r + 1.0 / 24.0;
    pl = pl * r + 1.0 / 6.0;
    pl = pl * r + 0.5;
    pl = pl * r + 1.0;
    pl = pl * r + 1.0;
    const long long k = (long long)kd;
    union { unsigned long long u; double d; } sc;
    sc.u = (unsigned long long)(k + 1023) << 52;
    return pl * sc.d;
}

// env.py:156-162 comp_pref_vel: unit vector from pos to goal in fp64; zero vector -> (1, 0).
// Positions are the simulator's fp32, targets are fp64 (Python floats in the reference).
CA_HD void pref_dir64(float px, float py, double gx, double gy, double* ox, double* oy) {
    const double dx = gx - (double)px;
    const double dy = gy - (double)py;
    const bool z = (dx == 0.0 && dy == 0.0);
    const double len = z ? 1.0 : sqrt(dx * dx + dy * dy);
    *ox = z ? 1.0 : dx / len;
    *oy = z ? 0.0 : dy / len;
}

// ---- Philox4x32-10 counter-based RNG ----
CA_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
CA_HD void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                      uint32_t* o) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}
enum { RNG_POS = 0, RNG_HEADING = 1, RNG_GOAL = 2, RNG_REGOAL = 3, RNG_RESET = 4, RNG_ALAN = 5 };
// two uniforms in [0,1) with 53 random bits each; stream = (seed, global arena, agent, purpose, seq)
CA_HD void rng2(uint64_t seed, int64_t arena, int agent, int purpose, uint32_t seq, double* u0, double* u1) {
    uint32_t w[4];
    const uint64_t g = (uint64_t)arena;
    philox4x32((uint32_t)g, (uint32_t)agent, (uint32_t)purpose, seq, (uint32_t)seed,
               (uint32_t)(seed >> 32) + (uint32_t)(g >> 32), w);
    const double k = 1.0 / 9007199254740992.0;
    *u0 = (double)(((uint64_t)(w[0] >> 5) << 26) | (uint64_t)(w[1] >> 6)) * k;
    *u1 = (double)(((uint64_t)(w[2] >> 5) << 26) | (uint64_t)(w[3] >> 6)) * k;
}
CA_HD double uniform64(double a, double b, double u) { return a + (b - a) * u; }

}  // namespace ca
