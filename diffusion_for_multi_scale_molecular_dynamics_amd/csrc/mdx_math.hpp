// MDX arithmetic for gfx950 device code.
//
// Fixed IEEE-754 operation sequences for log / exp / sinpi / cospi, the Philox4x32-10 counter RNG and the
// uint32 -> float conversions used by every kernel in this directory.  The sequences are specified in
// DESIGN.md ("MDX arithmetic"); the translation unit must be compiled with -ffp-contract=off so that the only
// fused operations are the explicit __builtin_fmaf / __builtin_fma calls.  Division and square root are the
// correctly rounded forms (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mdx {

__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }

// natural log, binary32: x = 2^k m, m in [sqrt(2)/2, sqrt(2)); f = m - 1; s = f / (2 + f)
__device__ __forceinline__ float logf_(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    int32_t ix = (int32_t)f2u(x);
    int32_t k = 0;
    if (ix < 0x00800000) {
        if ((ix & 0x7fffffff) == 0) return -__builtin_huge_valf();
        if (ix < 0) return __builtin_nanf("");
        k -= 25;
        x = x * two25;
        ix = (int32_t)f2u(x);
    }
    if (ix >= 0x7f800000) return x + x;
    k += (ix >> 23) - 127;
    ix &= 0x007fffff;
    const int32_t i = (ix + (0x95f64 << 3)) & 0x800000;
    x = u2f((uint32_t)(ix | (i ^ 0x3f800000)));
    k += (i >> 23);
    const float f = x - 1.0f;
    const float s = f / (2.0f + f);
    const float dk = (float)k;
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (Lg2 + w * Lg4);
    const float t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp, binary32: x = k ln2 + r; exp(r) = 1 + r + r c / (2 - c)
// The sequence of DESIGN.md's arithmetic contract (fdlibm's branches) evaluated WITHOUT divergent branches -- lanes of a wavefront
// hold arguments of every magnitude (SiLU over 64 neurons), so each branch of the original would be executed by
// every wavefront.  Equivalences used, all exact in IEEE-754:
//   * |x| < 1.5 ln2: the original sets k = +-1, hi = x -+ ln2HI, lo = +-ln2LO; the general formulas
//     hi = x - t ln2HI, lo = t ln2LO give the same bits for t = +-1 (t ln2HI is exact, x - (-a) == x + a);
//   * k = 0: 1 - ((x c)/(c - 2) - x) == 1 - (-(x c)/(2 - c) - x), a/(-b) == -(a/b): one division serves both forms.
// Special cases are selected at the end in the original's order of precedence.
__device__ __forceinline__ float expf_(float x)
{
    const float o_threshold = 8.8721679688e+01f, u_threshold = -1.0397208405e+02f;
    const float ln2HI = 6.9314575195e-01f, ln2LO = 1.4286067653e-06f, invln2 = 1.4426950216e+00f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    const uint32_t ux = f2u(x);
    const bool neg = (ux >> 31) != 0;
    const uint32_t hx = ux & 0x7fffffff;
    const bool reduce = hx > 0x3eb17218;                          // |x| > 0.5 ln2
    const bool unit = hx < 0x3F851592;                            // ... and |x| < 1.5 ln2: k = +-1
    const int32_t k_general = (int32_t)(invln2 * x + (neg ? -0.5f : 0.5f));
    const int32_t k = reduce ? (unit ? (neg ? -1 : 1) : k_general) : 0;
    const float t = (float)k;
    const float hi = x - t * ln2HI;
    const float lo = t * ln2LO;
    const float r = reduce ? hi - lo : x;
    const float tt = r * r;
    const float c = r - tt * (P1 + tt * P2);
    const float q = (r * c) / (2.0f - c);
    const float y0 = 1.0f - (-q - r);                             // k == 0
    const float y = 1.0f - ((lo - q) - hi);                       // k != 0
    const float scaled = (k >= -125) ? ((k == 128) ? (y * 2.0f) * 1.7014118346e+38f : y * u2f((uint32_t)(0x7f + k) << 23))
                                     : (y * u2f((uint32_t)(0x7f + (k + 100)) << 23)) * 7.8886090522e-31f;
    float out = (k == 0) ? y0 : scaled;
    if (!reduce && hx < 0x39000000) out = 1.0f + x;               // |x| < 2^-13
    if (x < u_threshold) out = 0.0f;
    if (x > o_threshold) out = __builtin_huge_valf();
    if (hx == 0x7f800000) out = neg ? 0.0f : x;
    if (hx > 0x7f800000) out = x + x;
    return out;
}

// natural log, binary64, finite normal positive arguments
__device__ __forceinline__ double log_(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    if (!(x > 0.0) || x > 1.7e308) return (x == 0.0) ? -__builtin_huge_val() : (x > 0.0 ? x : __builtin_nan(""));
    if (x < 2.2250738585072014e-308) return __builtin_nan("");
    const uint64_t ux = (uint64_t)__double_as_longlong(x);
    int64_t k = (int64_t)(ux >> 52) - 1023;
    const uint64_t m = ux & 0x000fffffffffffffULL;
    const uint64_t i = (m + 0x95f6400000000ULL) & 0x10000000000000ULL;
    const double xm = __longlong_as_double((long long)(m | (i ^ 0x3ff0000000000000ULL)));
    k += (int64_t)(i >> 52);
    const double f = xm - 1.0;
    const double s = f / (2.0 + f);
    const double dk = (double)k;
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    const double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    const double R = t2 + t1;
    const double hfsq = (0.5 * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// exp, binary64, |x| < 700
__device__ __forceinline__ double exp_(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (!(x > -700.0 && x < 700.0)) return (x >= 700.0) ? __builtin_huge_val() : (x <= -700.0 ? 0.0 : __builtin_nan(""));
    const int64_t k = (int64_t)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    const double t = (double)k;
    const double hi = x - t * ln2HI;
    const double lo = t * ln2LO;
    const double r = hi - lo;
    const double tt = r * r;
    const double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    const double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return y * __longlong_as_double((long long)((uint64_t)(1023 + k) << 52));
}

// sin(pi v), cos(pi v), v in [0, 2]
__device__ __forceinline__ void sincospif_(float v, float& s_out, float& c_out)
{
    const float S1 = 3.14159274f, S3 = -5.16771278f, S5 = 2.55016404f, S7 = -0.599264529f, S9 = 0.0821458866f;
    const float C2 = -4.93480220f, C4 = 4.05871213f, C6 = -1.33526277f, C8 = 0.235330630f, C10 = -0.0258068913f;
    const float q = __builtin_rintf(v * 2.0f);
    const float y = v - 0.5f * q;
    const float y2 = y * y;
    float ps = __builtin_fmaf(y2, S9, S7);
    ps = __builtin_fmaf(y2, ps, S5);
    ps = __builtin_fmaf(y2, ps, S3);
    ps = __builtin_fmaf(y2, ps, S1);
    const float sp = y * ps;
    float pc = __builtin_fmaf(y2, C10, C8);
    pc = __builtin_fmaf(y2, pc, C6);
    pc = __builtin_fmaf(y2, pc, C4);
    pc = __builtin_fmaf(y2, pc, C2);
    const float cp = __builtin_fmaf(y2, pc, 1.0f);
    const int qi = ((int)q) & 3;
    if (qi == 0) { s_out = sp; c_out = cp; }
    else if (qi == 1) { s_out = cp; c_out = -sp; }
    else if (qi == 2) { s_out = -sp; c_out = -cp; }
    else { s_out = -cp; c_out = sp; }
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw 2011)
struct u32x4 { uint32_t v[4]; };

__device__ __forceinline__ u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u32x4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// odd multiples of 2^-24 in (0,1)
__device__ __forceinline__ float u01(uint32_t r)
{
    return (float)(r >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f;
}

__device__ __forceinline__ void box_muller(uint32_t ra, uint32_t rb, float& z0, float& z1)
{
    const float u1 = u01(ra), u2 = u01(rb);
    const float rad = __builtin_sqrtf(-2.0f * logf_(u1));
    float s, c;
    sincospif_(2.0f * u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

__device__ __forceinline__ float gumbel_from_u(float u) { return -logf_(-logf_(u)); }

// y - floor(y), with 1.0 mapped to 0.0 (utils/basis_transformations.py:117-118)
__device__ __forceinline__ float wrap01(float y)
{
    float r = y - __builtin_floorf(y);
    return (r == 1.0f) ? 0.0f : r;
}

}  // namespace mdx
