/*
 * mdx_oracle.c -- CPU restatement of the reference's sampling hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle for the HIP kernels in
 * diffusion_for_multi_scale_molecular_dynamics_amd/csrc/.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product never does.
 *
 * It restates, in plain scalar C, the arithmetic of the reference (paths relative to
 * /root/reference/src/diffusion_for_multi_scale_molecular_dynamics/), each function citing the lines it follows.
 * Pinning: tests/test_oracle_golden.py checks every function here against the .npz fixtures in tests/golden, which were
 * produced by importing the reference itself (tests/golden/make_golden.py).
 *
 * Arithmetic contract ("MDX arithmetic"): every float operation below is an IEEE-754 binary32 (or, where a
 * variable is declared double, binary64) operation rounded to nearest-even, evaluated in exactly the order
 * written.  The file must be compiled with -ffp-contract=off; the only fused operations are the explicit
 * fmaf()/fma() calls.  log/exp/sinpi/cospi are NOT taken from libm: they are the fixed operation sequences
 * mdxo_logf/mdxo_expf/mdxo_log/mdxo_exp/mdxo_sincospif below (classic fdlibm-style argument reduction +
 * polynomial), so that a GPU kernel performing the same sequence produces the same bits.  That is what makes
 * "bit-exact atom-type draws" a property by construction rather than by luck.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MDXO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------------------ */
/* bit helpers                                                                                                  */
/* ------------------------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint64_t d2u(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
static inline double u2d(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }

/* ------------------------------------------------------------------------------------------------------------ */
/* MDX arithmetic: transcendental sequences                                                                     */
/* ------------------------------------------------------------------------------------------------------------ */

/* natural log, binary32.  x = 2^k * m, m in [sqrt(2)/2, sqrt(2)); f = m-1; s = f/(2+f);
 * log(m) = f - hfsq + s*(hfsq + R(s^2)).  <1 ulp. */
MDXO_API float mdxo_logf(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f, two25 = 3.355443200e+07f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    int32_t ix = (int32_t)f2u(x);
    int32_t k = 0;
    if (ix < 0x00800000) {                 /* x < 2^-126, zero, or negative */
        if ((ix & 0x7fffffff) == 0) return -INFINITY;
        if (ix < 0) return NAN;
        k -= 25;
        x = x * two25;
        ix = (int32_t)f2u(x);
    }
    if (ix >= 0x7f800000) return x + x;    /* inf or nan */
    k += (ix >> 23) - 127;
    ix &= 0x007fffff;
    int32_t i = (ix + (0x95f64 << 3)) & 0x800000;
    x = u2f((uint32_t)(ix | (i ^ 0x3f800000)));
    k += (i >> 23);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float dk = (float)k;
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp, binary32.  x = k ln2 + r, |r| <= ln2/2; exp(r) = 1 + r + r*c/(2-c), c = r - r^2 (P1 + r^2 P2). <1 ulp. */
MDXO_API float mdxo_expf(float x)
{
    const float o_threshold = 8.8721679688e+01f, u_threshold = -1.0397208405e+02f;
    const float ln2HI = 6.9314575195e-01f, ln2LO = 1.4286067653e-06f, invln2 = 1.4426950216e+00f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    uint32_t hx = f2u(x);
    int xsb = (int)(hx >> 31);
    hx &= 0x7fffffff;
    if (hx > 0x7f800000) return x + x;                    /* nan */
    if (hx == 0x7f800000) return xsb ? 0.0f : x;          /* +-inf */
    if (x > o_threshold) return INFINITY;
    if (x < u_threshold) return 0.0f;
    float hi = 0.0f, lo = 0.0f;
    int32_t k = 0;
    if (hx > 0x3eb17218) {                                /* |x| > 0.5 ln2 */
        if (hx < 0x3F851592) {                            /* |x| < 1.5 ln2 */
            if (xsb) { hi = x + ln2HI; lo = -ln2LO; k = -1; }
            else     { hi = x - ln2HI; lo = ln2LO;  k = 1; }
        } else {
            k = (int32_t)(invln2 * x + (xsb ? -0.5f : 0.5f));
            float t = (float)k;
            hi = x - t * ln2HI;
            lo = t * ln2LO;
        }
        x = hi - lo;
    } else if (hx < 0x39000000) {                         /* |x| < 2^-13 */
        return 1.0f + x;
    }
    float t = x * x;
    float c = x - t * (P1 + t * P2);
    if (k == 0) return 1.0f - ((x * c) / (c - 2.0f) - x);
    float y = 1.0f - ((lo - (x * c) / (2.0f - c)) - hi);
    if (k >= -125) {
        if (k == 128) return (y * 2.0f) * 1.7014118346e+38f;
        return y * u2f((uint32_t)(0x7f + k) << 23);
    }
    return (y * u2f((uint32_t)(0x7f + (k + 100)) << 23)) * 7.8886090522e-31f; /* 2^-100 */
}

/* natural log, binary64 (used only to form sigma_min*(sigma_max/sigma_min)^t in the schedule). */
MDXO_API double mdxo_log(double x)
{
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                 Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    /* domain here is a finite normal positive number; anything else is passed through conservatively */
    if (!(x > 0.0) || x > 1.7e308) return (x == 0.0) ? -INFINITY : (x > 0.0 ? x : NAN);
    if (x < 2.2250738585072014e-308) return NAN; /* subnormals never occur on this path */
    uint64_t ux = d2u(x);
    int64_t k = (int64_t)(ux >> 52) - 1023;
    uint64_t m = ux & 0x000fffffffffffffULL;
    /* choose m in [sqrt(2)/2, sqrt(2)) : if mantissa >= sqrt(2)-1 use x/2 */
    uint64_t i = (m + 0x95f6400000000ULL) & 0x10000000000000ULL;
    double xm = u2d(m | (i ^ 0x3ff0000000000000ULL));
    k += (int64_t)(i >> 52);
    double f = xm - 1.0;
    double s = f / (2.0 + f);
    double dk = (double)k;
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = (0.5 * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp, binary64, for |x| < 700. */
MDXO_API double mdxo_exp(double x)
{
    const double ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                 P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (!(x > -700.0 && x < 700.0)) return (x >= 700.0) ? INFINITY : (x <= -700.0 ? 0.0 : NAN);
    int64_t k = (int64_t)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
    double t = (double)k;
    double hi = x - t * ln2HI;
    double lo = t * ln2LO;
    double r = hi - lo;
    double tt = r * r;
    double c = r - tt * (P1 + tt * (P2 + tt * (P3 + tt * (P4 + tt * P5))));
    double y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
    return y * u2d((uint64_t)(1023 + k) << 52);
}

/* sin(pi v), cos(pi v) for v in [0, 2], binary32, by quadrant reduction (exact) and fmaf-Horner polynomials. */
MDXO_API void mdxo_sincospif(float v, float* s_out, float* c_out)
{
    const float S1 = 3.14159274f, S3 = -5.16771278f, S5 = 2.55016404f, S7 = -0.599264529f, S9 = 0.0821458866f;
    const float C2 = -4.93480220f, C4 = 4.05871213f, C6 = -1.33526277f, C8 = 0.235330630f, C10 = -0.0258068913f;
    float q = rintf(v * 2.0f);
    float y = v - 0.5f * q;
    float y2 = y * y;
    float ps = fmaf(y2, S9, S7);
    ps = fmaf(y2, ps, S5);
    ps = fmaf(y2, ps, S3);
    ps = fmaf(y2, ps, S1);
    float sp = y * ps;
    float pc = fmaf(y2, C10, C8);
    pc = fmaf(y2, pc, C6);
    pc = fmaf(y2, pc, C4);
    pc = fmaf(y2, pc, C2);
    float cp = fmaf(y2, pc, 1.0f);
    int qi = ((int)q) & 3;
    float s, c;
    if (qi == 0) { s = sp; c = cp; }
    else if (qi == 1) { s = cp; c = -sp; }
    else if (qi == 2) { s = -sp; c = -cp; }
    else { s = -cp; c = sp; }
    *s_out = s;
    *c_out = c;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* device-RNG specification: Philox4x32-10 (Salmon et al. 2011, public algorithm) + fixed float conversions     */
/* ------------------------------------------------------------------------------------------------------------ */
/* counter = (item, (call<<8)|sub, draw, tag); key = (seed_lo, seed_hi).
 * tags: */
enum { MDXO_TAG_COORD = 0, MDXO_TAG_GUMBEL = 1, MDXO_TAG_LATTICE = 2, MDXO_TAG_INIT = 3, MDXO_TAG_REPAINT_X0 = 4,
       MDXO_TAG_BINARY = 5, MDXO_TAG_REPAINT_Z = 6, MDXO_TAG_REPAINT_U = 7, MDXO_TAG_INIT_LATTICE = 8 };

MDXO_API void mdxo_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4])
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* uniform in (0,1): odd multiples of 2^-24 -- exact in binary32 */
static inline float mdxo_u01(uint32_t r) { return (float)(r >> 9) * 1.1920928955078125e-07f + 5.9604644775390625e-08f; }

static inline void mdxo_box_muller(uint32_t ra, uint32_t rb, float* z0, float* z1)
{
    float u1 = mdxo_u01(ra), u2 = mdxo_u01(rb);
    float rad = sqrtf(-2.0f * mdxo_logf(u1));
    float s, c;
    mdxo_sincospif(2.0f * u2, &s, &c);
    *z0 = rad * c;
    *z1 = rad * s;
}

/* n_items x width standard normals: value (item, k) = normal lane k%4 of call (k/4). */
MDXO_API void mdxo_rng_normal(uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag, int64_t n_items, int width,
                              float* out)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t it = 0; it < n_items; ++it)
        for (int sub = 0; sub * 4 < width; ++sub) {
            uint32_t r[4];
            float z[4];
            mdxo_philox4x32_10((uint32_t)it, (call << 8) | (uint32_t)sub, draw, tag, k0, k1, r);
            mdxo_box_muller(r[0], r[1], &z[0], &z[1]);
            mdxo_box_muller(r[2], r[3], &z[2], &z[3]);
            for (int l = 0; l < 4 && sub * 4 + l < width; ++l) out[it * width + sub * 4 + l] = z[l];
        }
}

/* n_items x width uniforms in (0,1) */
MDXO_API void mdxo_rng_uniform(uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag, int64_t n_items, int width,
                               float* out)
{
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t it = 0; it < n_items; ++it)
        for (int sub = 0; sub * 4 < width; ++sub) {
            uint32_t r[4];
            mdxo_philox4x32_10((uint32_t)it, (call << 8) | (uint32_t)sub, draw, tag, k0, k1, r);
            for (int l = 0; l < 4 && sub * 4 + l < width; ++l) out[it * width + sub * 4 + l] = mdxo_u01(r[l]);
        }
}

/* Gumbel(0,1) = -log(-log(u)) from the same uniforms (langevin_generator.py:100-107; the clip at small_epsilon
 * is a no-op because u >= 2^-24 > small_epsilon). */
MDXO_API void mdxo_rng_gumbel(uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag, int64_t n_items, int width,
                              float* out)
{
    mdxo_rng_uniform(seed, call, draw, tag, n_items, width, out);
    for (int64_t i = 0; i < n_items * width; ++i) out[i] = -mdxo_logf(-mdxo_logf(out[i]));
}

/* ------------------------------------------------------------------------------------------------------------ */
/* S1: variance-exploding schedule tables                                                                       */
/* noise_schedulers/noise_scheduler.py:112-267, sigma_calculator.py:72-74,102-104                               */
/* ------------------------------------------------------------------------------------------------------------ */
/* schedule_type: 0 exponential, 1 linear.  All outputs length T (matrices T*C*C), binary32.
 * The scalar hyper-parameters arrive as doubles (Python floats) and are narrowed exactly where torch narrows. */
MDXO_API int mdxo_noise_schedule(int T, int schedule_type, double time_delta, double sigma_min_d, double sigma_max_d,
                                 double corrector_eps, int C, float* time, float* sigma, float* sigma2, float* g,
                                 float* g2, float* eps, float* sqrt2eps, float* beta, float* alpha_bar, float* q,
                                 float* qbar, float* qbar_tm1)
{
    if (T < 2 || C < 2 || (schedule_type != 0 && schedule_type != 1)) return -1;
    const float start = (float)time_delta, end = 1.0f;
    const float step = (end - start) / (float)(T - 1);      /* torch.linspace (noise_scheduler.py:186-189) */
    const int halfway = T / 2;
    const float smin = (float)sigma_min_d, smax = (float)sigma_max_d;
    const float ratio = smax / smin;                         /* sigma_calculator.py:66-68 */
    const float diff = smax - smin;                          /* sigma_calculator.py:96-98 */
    const double log_ratio = mdxo_log((double)ratio);
    /* torch.linspace (aten RangeFactoriesKernel.cpp): start + step*i below the midpoint, end - step*(T-1-i) above,
     * each contracted to one fused multiply-add in the shipped binary (pinned by tests/golden/schedules.npz). */
    for (int i = 0; i < T; ++i) {
        float t = (i < halfway) ? fmaf(step, (float)i, start) : fmaf(-step, (float)(T - 1 - i), end);
        time[i] = t;
        float s;
        if (schedule_type == 0) {
            float p = (float)mdxo_exp((double)t * log_ratio);   /* ratio ** t, correctly rounded to binary32 */
            s = smin * p;                                        /* sigma_calculator.py:72-74 */
        } else {
            s = smin + diff * t;                                 /* sigma_calculator.py:102-104 */
        }
        sigma[i] = s;
        sigma2[i] = s * s;
    }
    /* g^2_i = sigma^2_i - sigma^2_{i-1}; first uses sigma_min^2 (Python double, narrowed) :191-199 */
    g2[0] = sigma2[0] - (float)(sigma_min_d * sigma_min_d);
    for (int i = 1; i < T; ++i) g2[i] = sigma2[i] - sigma2[i - 1];
    for (int i = 0; i < T; ++i) g[i] = sqrtf(g2[i]);
    /* epsilon :201-218 */
    /* python_double / tensor is Tensor.__rtruediv__ = reciprocal(tensor) * scalar */
    eps[0] = (1.0f / sigma2[0]) * (float)(0.5 * corrector_eps * (sigma_min_d * sigma_min_d));
    const float half_eps = (float)(0.5 * corrector_eps);
    for (int i = 1; i < T; ++i) eps[i] = (half_eps * sigma2[i - 1]) / sigma2[0];
    for (int i = 0; i < T; ++i) sqrt2eps[i] = sqrtf(2.0f * eps[i]);
    /* beta_t = 1/(T - t + 1), t = 1..T :220-222 ; alpha_bar = cumprod(1-beta) :224-226 */
    double ab = 1.0;                                       /* torch.cumprod accumulates binary32 in binary64 on CPU */
    for (int i = 0; i < T; ++i) {
        beta[i] = 1.0f / (float)(T - i);
        ab = ab * (double)(1.0f - beta[i]);
        alpha_bar[i] = (float)ab;
    }
    /* Q_t = (1-beta) I + beta 1 e_MASK^T :228-242 */
    const int M = C - 1;
    for (int i = 0; i < T; ++i) {
        float omb = 1.0f - beta[i];
        for (int r = 0; r < C; ++r)
            for (int c = 0; c < C; ++c) {
                float v = omb * (r == c ? 1.0f : 0.0f);
                v = v + beta[i] * (c == M ? 1.0f : 0.0f);
                q[(i * C + r) * C + c] = v;
            }
    }
    /* Qbar_t = Qbar_{t-1} Q_t, sequential binary32 matmul :244-252 ; Qbar_{t-1} with identity first :254-267 */
    for (int r = 0; r < C; ++r)
        for (int c = 0; c < C; ++c) {
            qbar[r * C + c] = q[r * C + c];
            qbar_tm1[r * C + c] = (r == c) ? 1.0f : 0.0f;
        }
    for (int i = 1; i < T; ++i) {
        const float* prev = qbar + (size_t)(i - 1) * C * C;
        const float* qt = q + (size_t)i * C * C;
        float* cur = qbar + (size_t)i * C * C;
        for (int r = 0; r < C; ++r)
            for (int c = 0; c < C; ++c) {
                float acc = 0.0f;
                for (int k = 0; k < C; ++k) acc = fmaf(prev[r * C + k], qt[k * C + c], acc);
                cur[r * C + c] = acc;
            }
        memcpy(qbar_tm1 + (size_t)i * C * C, prev, sizeof(float) * C * C);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* P1: relative-coordinates update + wrap                                                                       */
/* generators/langevin_generator.py:194-201 ; utils/basis_transformations.py:117-118                            */
/* ------------------------------------------------------------------------------------------------------------ */
static inline float mdxo_wrap01(float y)
{
    float r = y - floorf(y);         /* == torch.remainder(y, 1.0) for finite y (wrap) */
    if (r == 1.0f) r = 0.0f;         /* basis_transformations.py:118 */
    return r;
}

MDXO_API void mdxo_wrap(const float* y, int64_t n, float* out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = mdxo_wrap01(y[i]);
}

/* x' = wrap((x + (w*s)/sigma) + n*z) */
MDXO_API void mdxo_coordinates_update(const float* x, const float* s, const float* z, float w, float n, float sigma,
                                      int64_t count, float* out)
{
    for (int64_t i = 0; i < count; ++i) {
        float a = (w * s[i]) / sigma;
        float b = n * z[i];
        out[i] = mdxo_wrap01((x[i] + a) + b);
    }
}

/* P3: l' = (l + (w*s)/sigma_n) + n*z   generators/langevin_generator.py:485-490 */
MDXO_API void mdxo_lattice_update(const float* l, const float* s, const float* z, float w, float n, float sigma_n,
                                  int64_t count, float* out)
{
    for (int64_t i = 0; i < count; ++i) out[i] = (l[i] + (w * s[i]) / sigma_n) + n * z[i];
}

/* ------------------------------------------------------------------------------------------------------------ */
/* P2: atom-type update                                                                                          */
/* generators/langevin_generator.py:247-439 ; utils/d3pm_utils.py:64-150                                       */
/* ------------------------------------------------------------------------------------------------------------ */
#define MDXO_MAXC 64

/* p(a_{t-1} | a_t, logits) for one atom.  d3pm_utils.py:127-150 (softmax, clip, renormalise) and :105-124 */
static void mdxo_posterior(const float* logits, int a_t, const float* q, const float* qbar, const float* qbar_tm1,
                           int C, float small_eps, float* p)
{
    float e[MDXO_MAXC];
    float m = logits[0];
    for (int c = 1; c < C; ++c) m = (logits[c] > m) ? logits[c] : m;
    float S = 0.0f;
    for (int c = 0; c < C; ++c) { e[c] = mdxo_expf(logits[c] - m); S = S + e[c]; }
    float S2 = 0.0f;
    const float invS = 1.0f / S;                          /* aten SoftMaxKernel: multiply by the reciprocal of the sum */
    for (int c = 0; c < C; ++c) {
        float r = e[c] * invS;
        r = (r < small_eps) ? small_eps : r;             /* .clip(min=eps) */
        e[c] = r;
        S2 = S2 + r;
    }
    for (int c = 0; c < C; ++c) e[c] = e[c] / S2;         /* gamma_0 */
    float den = 0.0f;
    for (int j = 0; j < C; ++j) den = den + e[j] * qbar[j * C + a_t];       /* gamma_0 . Qbar_t . a_t */
    for (int i = 0; i < C; ++i) {
        float num1 = 0.0f;
        for (int j = 0; j < C; ++j) num1 = num1 + e[j] * qbar_tm1[j * C + i];   /* gamma_0 . Qbar_{t-1} */
        float num2 = q[i * C + a_t];                                             /* Q_t . a_t            */
        p[i] = (num1 * num2) / den;
    }
}

/* Full update for a batch.  gumbel [B,N,C]; u [B,N] (only read if greedy).  p_out (nullable) receives the
 * probabilities after the greedy adjustment, gumbel_out (nullable) the Gumbel values actually used. */
MDXO_API int mdxo_atom_types_update(const float* logits, const int64_t* a, const float* q, const float* qbar,
                                    const float* qbar_tm1, const float* gumbel, const float* u, int64_t B, int N,
                                    int C, float small_eps, int greedy, int one_transition, int64_t* a_out,
                                    float* p_out, float* gumbel_out)
{
    if (C > MDXO_MAXC || C < 2) return -1;
    const int M = C - 1;
    float* vmax = (float*)malloc(sizeof(float) * (size_t)N);
    int64_t* prop = (int64_t*)malloc(sizeof(int64_t) * (size_t)N);
    for (int64_t b = 0; b < B; ++b) {
        int all_masked = 1;                                           /* :406-408 */
        for (int n = 0; n < N; ++n) all_masked &= (a[b * N + n] == M);
        for (int n = 0; n < N; ++n) {
            const int64_t at = b * N + n;
            float p[MDXO_MAXC], gm[MDXO_MAXC];
            int a_t = (int)a[at];
            mdxo_posterior(logits + at * C, a_t, q, qbar, qbar_tm1, C, small_eps, p);
            for (int c = 0; c < C; ++c) gm[c] = gumbel[at * C + c];
            if (greedy) {                                             /* :382-439 */
                int unmask = u[at] > p[M];
                if (!all_masked && unmask && a_t == M) p[M] = 0.0f;
                if (!all_masked) for (int c = 0; c < C; ++c) gm[c] = 0.0f;
            }
            float best = 0.0f; int arg = 0;                         /* torch.max: first maximal index */
            for (int c = 0; c < C; ++c) {
                float v = mdxo_logf(p[c] + small_eps) + gm[c];       /* :311-315 */
                if (c == 0 || v > best) { best = v; arg = c; }
                if (p_out) p_out[at * C + c] = p[c];
                if (gumbel_out) gumbel_out[at * C + c] = gm[c];
            }
            vmax[n] = best;
            prop[n] = arg;
        }
        if (one_transition) {                                         /* :339-380 */
            int k = 0;
            float bestv = -INFINITY;
            for (int n = 0; n < N; ++n) {
                float v = (prop[n] != a[b * N + n]) ? vmax[n] : -INFINITY;
                if (n == 0 || v > bestv) { bestv = v; k = n; }
            }
            for (int n = 0; n < N; ++n) a_out[b * N + n] = a[b * N + n];
            a_out[b * N + k] = prop[k];
        } else {
            for (int n = 0; n < N; ++n) a_out[b * N + n] = prop[n];
        }
    }
    free(vmax);
    free(prop);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* F1 / F2: forward noising (used by repaint)                                                                   */
/* noisers/relative_coordinates_noiser.py:33-67 ; noisers/atom_types_noiser.py:30-60 ; d3pm_utils.py:23-39     */
/* ------------------------------------------------------------------------------------------------------------ */
MDXO_API void mdxo_noise_coordinates(const float* x0, const float* z, float sigma, int64_t count, float* out)
{
    for (int64_t i = 0; i < count; ++i) out[i] = mdxo_wrap01(x0[i] + sigma * z[i]);
}

/* a_t = argmax_c( log(Qbar[a0][c]) + (-log(-log u_c)) ); u is NOT clipped (quirk 6) */
MDXO_API void mdxo_noise_atom_types(const int64_t* a0, const float* qbar, const float* u, int64_t n_atoms, int C,
                                    int64_t* out)
{
    for (int64_t i = 0; i < n_atoms; ++i) {
        float best = 0.0f;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            float lq = mdxo_logf(qbar[a0[i] * C + c]);
            float gn = -mdxo_logf(-mdxo_logf(u[i * C + c]));
            float v = lq + gn;
            /* torch.argmax: first maximal index; a NaN is treated as maximal */
            if (c == 0 || v > best || (v != v && best == best)) { best = v; arg = c; }
        }
        out[i] = arg;
    }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* N1: periodic radius graph, 27 images                                                                         */
/* utils/neighbors.py:36-224 ; utils/lattice_utils.py:10-29 ; models/egnn_utils.py:107-144                      */
/* ------------------------------------------------------------------------------------------------------------ */
/* lattice vector of image l (itertools.product(-1,0,1)^3 order) = rel @ cell, binary32, sequential k */
static void mdxo_image_vectors(const float* cell, float lv[27][3])
{
    int l = 0;
    for (int i0 = -1; i0 <= 1; ++i0)
        for (int i1 = -1; i1 <= 1; ++i1)
            for (int i2 = -1; i2 <= 1; ++i2, ++l) {
                float rel[3] = {(float)i0, (float)i1, (float)i2};
                for (int c = 0; c < 3; ++c) {
                    float acc = 0.0f;
                    for (int k = 0; k < 3; ++k) acc = fmaf(rel[k], cell[k * 3 + c], acc);
                    lv[l][c] = acc;
                }
            }
}

/* shortest distance that crosses the unit cell (neighbors.py:323-351), binary32 */
MDXO_API float mdxo_shortest_crossing_distance(const float* cell)
{
    const float* a1 = cell; const float* a2 = cell + 3; const float* a3 = cell + 6;
    float c12[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2], a1[0] * a2[1] - a1[1] * a2[0]};
    float c13[3] = {a1[1] * a3[2] - a1[2] * a3[1], a1[2] * a3[0] - a1[0] * a3[2], a1[0] * a3[1] - a1[1] * a3[0]};
    float c23[3] = {a2[1] * a3[2] - a2[2] * a3[1], a2[2] * a3[0] - a2[0] * a3[2], a2[0] * a3[1] - a2[1] * a3[0]};
    float vol = fabsf((c12[0] * a3[0] + c12[1] * a3[1]) + c12[2] * a3[2]);
    float n12 = sqrtf((c12[0] * c12[0] + c12[1] * c12[1]) + c12[2] * c12[2]);
    float n13 = sqrtf((c13[0] * c13[0] + c13[1] * c13[1]) + c13[2] * c13[2]);
    float n23 = sqrtf((c23[0] * c23[0] + c23[1] * c23[1]) + c23[2] * c23[2]);
    float d = vol / n12;
    float d2 = vol / n13; if (d2 < d) d = d2;
    float d3 = vol / n23; if (d3 < d) d = d3;
    return d;
}

static inline int mdxo_pair_image_valid(const float* pi, const float* pj, const float* lv, float rc2)
{
    float sx = pj[0] + lv[0], sy = pj[1] + lv[1], sz = pj[2] + lv[2];   /* neighbors.py:241-244 */
    float dx = pi[0] - sx, dy = pi[1] - sy, dz = pi[2] - sz;
    float d2 = (dx * dx + dy * dy) + dz * dz;                            /* neighbors.py:175 */
    return (0.0f < d2) && (d2 <= rc2);                                   /* neighbors.py:192-194 */
}

/* mode 0: every (b, i, j, image) edge, ordered by (b, i, j, image)   [full adjacency info]
 * mode 1: unique (b, i, j) pairs, ordered by (b, i, j)               [EGNN: torch.unique(dim=1)]
 * Two-call protocol: with src == NULL only counts are produced.  Returns total edge count, or -2 if the cutoff
 * reaches beyond the first image shell for some structure (neighbors.py:107-113).
 * counts: [B*N] edges per source atom; src/dst int64 per-structure indices (mode 0) or batch-global (mode 1);
 * image: [E] image index 0..26 (mode 0 only; shifts = image vector). */
MDXO_API int64_t mdxo_radius_graph(const float* cart, const float* cell, float rc, int64_t B, int N, int mode,
                                   int64_t* counts, int64_t* src, int64_t* dst, int32_t* image)
{
    const float rc2 = rc * rc;
    int64_t E = 0;
    for (int64_t b = 0; b < B; ++b) {
        if (!(mdxo_shortest_crossing_distance(cell + b * 9) > rc)) return -2;
        float lv[27][3];
        mdxo_image_vectors(cell + b * 9, lv);
        const float* P = cart + b * N * 3;
        for (int i = 0; i < N; ++i) {
            int64_t cnt = 0;
            for (int j = 0; j < N; ++j) {
                int any = 0;
                for (int l = 0; l < 27; ++l) {
                    if (!mdxo_pair_image_valid(P + i * 3, P + j * 3, lv[l], rc2)) continue;
                    if (mode == 0) {
                        if (src) { src[E] = i; dst[E] = j; image[E] = l; }
                        ++E; ++cnt;
                    } else any = 1;
                }
                if (mode == 1 && any) {
                    if (src) { src[E] = b * N + i; dst[E] = b * N + j; }
                    ++E; ++cnt;
                }
            }
            if (counts) counts[b * N + i] = cnt;
        }
    }
    return E;
}

MDXO_API void mdxo_image_vectors_out(const float* cell, float* out27x3)
{
    float lv[27][3];
    mdxo_image_vectors(cell, lv);
    memcpy(out27x3, lv, sizeof(lv));
}
