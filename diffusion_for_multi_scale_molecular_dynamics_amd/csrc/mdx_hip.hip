// HIP kernels (gfx950 / CDNA4) + C ABI of the sampling hot path.  See include/mdx_hip.h for the contract and
// DESIGN.md for data layout, roofline accounting and the arithmetic specification.
//
// Kernel inventory
//   schedule_kernel        S1  variance-exploding schedule tables (one-off, one workgroup)
//   fill_time_sigma_kernel     TIME / NOISE [B,1] network inputs from the device tables
//   coords_update_kernel   P1  x' = wrap((x + w s / sigma) + n z)           flat, 16 B/lane
//   pc_step_kernel         P2 (+P1 +P3) fused per-step update, one lane-group per structure,
//                              wavefront-shuffle reductions over the atoms of a structure
//   repaint_rows_kernel    R1  forward-noise + scatter of the constrained rows (F1 + F2 fused)
//   radius_graph_kernel    N1  27-image radius graph, structure tile staged in LDS, one wavefront per source
//                              row, ballot/scan ranked writes => edges come out sorted, no atomics
//   rng_fill_kernel            Philox draws as arrays
// 64-wide wavefronts are assumed throughout (gfx950).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cmath>
#include <cstdlib>

#include "../../include/mdx_hip.h"
#include "mdx_math.hpp"

using namespace mdx;

namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;

// the call index of a counter-based RNG request: the device word when the request names one (launches captured into a hipGraph)
__device__ __forceinline__ uint32_t rng_call(const mdx_rng_t& r) { return r.call_dev ? *r.call_dev : r.call; }


inline int launch_status() { return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP; }
inline hipStream_t as_stream(mdx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------------------------
// S1
// ---------------------------------------------------------------------------------------------------------------
struct ScheduleArgs {
    int T, type, C;
    double time_delta, sigma_min, sigma_max, corrector_eps;
    float *time, *sigma, *sigma2, *g, *g2, *eps, *sqrt2eps, *beta, *alpha_bar, *q, *qbar, *qbar_tm1;
};

// Row r of Qbar_t = Qbar_{t-1} Q_t for t = 0..T-1 (row r of the product depends only on row r of Qbar_{t-1}: the C
// rows are independent chains, one lane each).  CSPEC > 0 substitutes the class count as a literal so that the class
// loops unroll to straight-line code -- with a runtime count every (c < C) predicate of this single-lane, T-step
// sequential loop is a branch (0.29 us per step); CSPEC = 0 is the generic form.  Same arithmetic.
template <int CSPEC>
__device__ __forceinline__ void qbar_chain(const ScheduleArgs& a, int r)
{
    const int T = a.T, C = CSPEC > 0 ? CSPEC : a.C;
#define MDX_SCHED_CLASSES(c) _Pragma("unroll") for (int c = 0; c < MDX_MAX_CLASSES; ++c) if (c < C)
    float prev[MDX_MAX_CLASSES], cur[MDX_MAX_CLASSES];
    MDX_SCHED_CLASSES(c) {
        const float b = 1.0f / (float)T;
        float v = (1.0f - b) * (r == c ? 1.0f : 0.0f);
        v = v + b * (c == C - 1 ? 1.0f : 0.0f);
        prev[c] = v;
        a.qbar[r * C + c] = v;
        a.qbar_tm1[r * C + c] = (r == c) ? 1.0f : 0.0f;
    }
#pragma unroll 4
    for (int i = 1; i < T; ++i) {                  // only the fmaf chain through prev[] is sequential
        const float b = 1.0f / (float)(T - i);
        const float omb = 1.0f - b;
        MDX_SCHED_CLASSES(c) {
            float acc = 0.0f;
            MDX_SCHED_CLASSES(k) {
                float qkc = omb * (k == c ? 1.0f : 0.0f);
                qkc = qkc + b * (c == C - 1 ? 1.0f : 0.0f);
                acc = __builtin_fmaf(prev[k], qkc, acc);
            }
            cur[c] = acc;
        }
        MDX_SCHED_CLASSES(c) {
            a.qbar_tm1[((int64_t)i * C + r) * C + c] = prev[c];
            a.qbar[((int64_t)i * C + r) * C + c] = cur[c];
            prev[c] = cur[c];
        }
    }
#undef MDX_SCHED_CLASSES
}

__global__ __launch_bounds__(kBlock) void schedule_kernel(ScheduleArgs a)
{
    const int T = a.T, C = a.C;
    const float start = (float)a.time_delta, end = 1.0f;
    const float step = (end - start) / (float)(T - 1);
    const int halfway = T / 2;
    const float smin = (float)a.sigma_min, smax = (float)a.sigma_max;
    const float ratio = smax / smin;
    const float diff = smax - smin;
    const double log_ratio = log_((double)ratio);
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float t = (i < halfway) ? __builtin_fmaf(step, (float)i, start)
                                      : __builtin_fmaf(-step, (float)(T - 1 - i), end);
        a.time[i] = t;
        float s;
        if (a.type == 0) {
            const float p = (float)exp_((double)t * log_ratio);
            s = smin * p;
        } else {
            s = smin + diff * t;
        }
        a.sigma[i] = s;
        a.sigma2[i] = s * s;
        const float b = 1.0f / (float)(T - i);
        a.beta[i] = b;
        const float omb = 1.0f - b;
        for (int r = 0; r < C; ++r)
            for (int c = 0; c < C; ++c) {
                float v = omb * (r == c ? 1.0f : 0.0f);
                v = v + b * (c == C - 1 ? 1.0f : 0.0f);
                a.q[((int64_t)i * C + r) * C + c] = v;
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float g2 = (i == 0) ? a.sigma2[0] - (float)(a.sigma_min * a.sigma_min) : a.sigma2[i] - a.sigma2[i - 1];
        a.g2[i] = g2;
        a.g[i] = __builtin_sqrtf(g2);
        float e;
        if (i == 0) e = (1.0f / a.sigma2[0]) * (float)(0.5 * a.corrector_eps * (a.sigma_min * a.sigma_min));
        else e = ((float)(0.5 * a.corrector_eps) * a.sigma2[i - 1]) / a.sigma2[0];
        a.eps[i] = e;
        a.sqrt2eps[i] = __builtin_sqrtf(2.0f * e);
    }
    // sequential chains: alpha_bar (binary64 running product) on one lane; Qbar row r on lane r (row r of
    // Qbar_t depends only on row r of Qbar_{t-1}, so rows are independent chains)
    if (threadIdx.x == kWave) {
        double ab = 1.0;
        // unrolled: the divisions of consecutive steps are independent of the running product and overlap
#pragma unroll 8
        for (int i = 0; i < T; ++i) {
            ab = ab * (double)(1.0f - 1.0f / (float)(T - i));
            a.alpha_bar[i] = (float)ab;
        }
    }
    if ((int)threadIdx.x < C) {
        if (C == 2) qbar_chain<2>(a, threadIdx.x);
        else if (C == 3) qbar_chain<3>(a, threadIdx.x);
        else qbar_chain<0>(a, threadIdx.x);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// step index on the device
// ---------------------------------------------------------------------------------------------------------------
__global__ void index_kernel(int32_t* d_index, int32_t value, int add)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *d_index = add ? *d_index + value : value;
}

struct SchedDev {  // by-value copy of mdx_schedule_t for kernels
    int T, C;
    double sigma_min;
    const float *time, *sigma, *g, *g2, *eps, *q, *qbar, *qbar_tm1;
};

inline SchedDev to_dev(const mdx_schedule_t* s)
{
    SchedDev d;
    d.T = s->total_time_steps;
    d.C = s->num_classes;
    d.sigma_min = s->sigma_min;
    d.time = s->time; d.sigma = s->sigma; d.g = s->g; d.g2 = s->g_squared; d.eps = s->epsilon;
    d.q = s->q_matrix; d.qbar = s->q_bar_matrix; d.qbar_tm1 = s->q_bar_tm1_matrix;
    return d;
}

struct StepScalars {
    float time, sigma, w, n, sigma_n;
    int idx;        // row of the Q tables
    int index;      // effective index_i
};

// predictor_step scalars (generators/langevin_generator.py:559-569) / corrector_step scalars (:719-733, :678, :749)
__device__ __forceinline__ StepScalars step_scalars(const SchedDev& s, int mode, int index, double atoms_pow)
{
    StepScalars o;
    o.index = index;
    if (mode == MDX_PREDICTOR) {
        const int idx = index - 1;
        o.idx = idx;
        o.time = s.time[idx];
        o.sigma = s.sigma[idx];
        o.w = s.g2[idx];
        o.n = s.g[idx];
        o.sigma_n = o.sigma / (float)atoms_pow;
    } else {
        if (index == 0) {
            o.idx = 0;
            o.time = 0.0f;
            o.sigma = (float)s.sigma_min;
            o.sigma_n = (float)(s.sigma_min / atoms_pow);
        } else {
            o.idx = index - 1;
            o.time = s.time[o.idx];
            o.sigma = s.sigma[o.idx];
            o.sigma_n = o.sigma / (float)atoms_pow;
        }
        o.w = s.eps[index];
        o.n = __builtin_sqrtf(2.0f * o.w);
    }
    return o;
}

__global__ __launch_bounds__(kBlock) void fill_time_sigma_kernel(SchedDev s, int mode, int index_i,
                                                                 const int32_t* d_index, float* time_out,
                                                                 float* sigma_out, int64_t batch)
{
    const int index = (d_index ? *d_index : 0) + index_i;
    const StepScalars sc = step_scalars(s, mode, index, 1.0);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < batch; i += (int64_t)gridDim.x * blockDim.x) {
        time_out[i] = sc.time;
        sigma_out[i] = sc.sigma;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// P1 / P3 / F1: flat elementwise kernels
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float coord_update(float x, float s, float z, float w, float n, float sigma)
{
    return wrap01((x + (w * s) / sigma) + n * z);
}

// VEC = 4: 16 B per lane per array; requires 16-B aligned pointers.  Tail handled by the scalar instantiation.
template <int VEC>
__global__ __launch_bounds__(kBlock) void coords_update_kernel(const float* __restrict__ x, const float* __restrict__ s,
                                                               const float* __restrict__ z, float w, float n,
                                                               float sigma, int64_t count, float* __restrict__ out,
                                                               const float* __restrict__ weights_dev)
{
    if (weights_dev) {                  // {score weight, noise weight, sigma} computed on the device (adaptive corrector)
        w = weights_dev[0];
        n = weights_dev[1];
        sigma = weights_dev[2];
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    if (VEC == 4) {
        const int64_t nvec = count >> 2;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const float4* s4 = reinterpret_cast<const float4*>(s);
        const float4* z4 = reinterpret_cast<const float4*>(z);
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += stride) {
            const float4 xv = x4[i], sv = s4[i], zv = z4[i];
            float4 o;
            o.x = coord_update(xv.x, sv.x, zv.x, w, n, sigma);
            o.y = coord_update(xv.y, sv.y, zv.y, w, n, sigma);
            o.z = coord_update(xv.z, sv.z, zv.z, w, n, sigma);
            o.w = coord_update(xv.w, sv.w, zv.w, w, n, sigma);
            o4[i] = o;
        }
        for (int64_t i = (nvec << 2) + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += stride)
            out[i] = coord_update(x[i], s[i], z[i], w, n, sigma);
    } else {
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += stride)
            out[i] = coord_update(x[i], s[i], z[i], w, n, sigma);
    }
}

__global__ __launch_bounds__(kBlock) void lattice_update_kernel(const float* l, const float* s, const float* z, float w,
                                                                float n, float sigma_n, int64_t count, float* out,
                                                                const float* weights_dev)
{
    if (weights_dev) {
        w = weights_dev[0];
        n = weights_dev[1];
        sigma_n = weights_dev[2];
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (l[i] + (w * s[i]) / sigma_n) + n * z[i];
}

__global__ __launch_bounds__(kBlock) void noise_coords_kernel(const float* x0, const float* z, float sigma, int64_t count,
                                                              float* out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = wrap01(x0[i] + sigma * z[i]);
}

// the reference's own signatures: one sigma per element (noisers/relative_coordinates_noiser.py:33-67, lattice_noiser.py:47-81)
__global__ __launch_bounds__(kBlock) void noise_coords_sigmas_kernel(const float* x0, const float* z, const float* sigmas,
                                                                     int64_t count, float* out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = wrap01(x0[i] + sigmas[i] * z[i]);
}

__global__ __launch_bounds__(kBlock) void noise_lattice_kernel(const float* l0, const float* z, const float* sigmas_n,
                                                               int64_t count, float* out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = sigmas_n[i] * z[i] + l0[i];
}

// ---------------------------------------------------------------------------------------------------------------
// P2 (+P1 +P3): fused per-step update
// ---------------------------------------------------------------------------------------------------------------
struct PcArgs {
    SchedDev sched;
    int use_tables;          // 1: scalars and Q matrices from the tables at the effective index; 0: explicit
    int mode, index_i;
    const int32_t* d_index;
    double atoms_pow;        // number_of_atoms ** (1/spatial_dimension), computed in binary64 like the reference
    // explicit-operand form (stand-alone P2)
    const float *q_explicit, *qbar_explicit, *qbar_tm1_explicit;
    // flags
    int greedy, one_transition, fixed_lattice, update_types, do_coords, do_lattice;
    float small_eps;
    // operands
    const int64_t* a;
    const float *x, *l, *logits, *score_x, *score_l;
    const float *z_coord, *gumbel, *u, *z_lat;
    mdx_rng_t rng;
    int64_t B;
    int N, d, C, nl;
    int64_t* a_out;
    float *x_out, *l_out, *p_out;
    uint32_t* status;
};

// posterior p(a_{t-1} | a_t, logits) for one atom (utils/d3pm_utils.py:105-150).  Loops run to the compile-time
// bound MDX_MAX_CLASSES with a (c < C) predicate so that e[] / p[] stay in registers (no scratch); the operations
// performed, and their order, are those of the runtime-bound loops.
#define MDX_FOR_CLASSES(c) _Pragma("unroll") for (int c = 0; c < MDX_MAX_CLASSES; ++c) if (c < C)
// softmax -> clip at small_epsilon -> renormalise: the class probabilities the posterior starts from
__device__ __forceinline__ void clipped_softmax(const float* lg, int C, float small_eps, float* e)
{
    float m = lg[0];
    MDX_FOR_CLASSES(c) if (c > 0) m = (lg[c] > m) ? lg[c] : m;
    float S = 0.0f;
    MDX_FOR_CLASSES(c) {
        e[c] = expf_(lg[c] - m);
        S = S + e[c];
    }
    const float invS = 1.0f / S;
    float S2 = 0.0f;
    MDX_FOR_CLASSES(c) {
        float r = e[c] * invS;
        r = (r < small_eps) ? small_eps : r;
        e[c] = r;
        S2 = S2 + r;
    }
    MDX_FOR_CLASSES(c) e[c] = e[c] / S2;
}

// With ONE atom type (C = 2) the logits are (l0, -inf) -- the network API forces the MASK logit to -inf
// (score_network.py:183-185) -- and clipped_softmax does not depend on l0 as long as it is finite: l0 - max = 0 exactly,
// exp(0) = 1, exp(-inf) = 0.  Its result is then a per-launch constant: the same operations on (0, -inf), evaluated once
// (the persistent sampler does this outside its loop) instead of two exp and three divisions per atom and step.
struct FixedSoftmaxC2 {
    float e0, e1;
    int valid;
};

__device__ __forceinline__ FixedSoftmaxC2 fixed_softmax_c2(float small_eps)
{
    constexpr int C = 2;
    float lg[MDX_MAX_CLASSES] = {0.0f, -__builtin_huge_valf()}, e[MDX_MAX_CLASSES];
    clipped_softmax(lg, C, small_eps, e);
    return FixedSoftmaxC2{e[0], e[1], 1};
}

__device__ __forceinline__ void posterior(const float* __restrict__ logits, int a_t, const float* __restrict__ q,
                                          const float* __restrict__ qbar, const float* __restrict__ qbar_tm1, int C,
                                          float small_eps, float* p, const FixedSoftmaxC2& fixed = FixedSoftmaxC2{0.0f, 0.0f, 0})
{
    float e[MDX_MAX_CLASSES], lg[MDX_MAX_CLASSES];
    MDX_FOR_CLASSES(c) lg[c] = logits[c];
    if (C == 2 && fixed.valid && lg[1] == -__builtin_huge_valf() && __builtin_fabsf(lg[0]) < __builtin_huge_valf()) {
        MDX_FOR_CLASSES(c) e[c] = c == 0 ? fixed.e0 : fixed.e1;
    } else {
        clipped_softmax(lg, C, small_eps, e);
    }
    float den = 0.0f;
    MDX_FOR_CLASSES(j) den = den + e[j] * qbar[j * C + a_t];
    MDX_FOR_CLASSES(i) {
        float num1 = 0.0f;
        MDX_FOR_CLASSES(j) num1 = num1 + e[j] * qbar_tm1[j * C + i];
        const float num2 = q[i * C + a_t];
        p[i] = (num1 * num2) / den;
    }
}

// The value of a partner lane at "distance" O inside a group of lanes, for all-to-all reductions over the group
// (AND, arg-max over a total order: any complete exchange pattern gives every lane the same result).  O = 1, 2: DPP
// quad permutes; O = 4, 8: DPP mirrors inside 8 / 16 lanes (lane i <-> 7-i / 15-i pairs the two halves once the halves
// are reduced); O = 16, 32: cross-row, through ds_bpermute.  The DPP forms are register moves; a ds_bpermute goes
// through the LDS crossbar and costs ~100 cycles of latency each on a lone wavefront.
template <int O>
__device__ __forceinline__ int partner(int v)
{
    if constexpr (O == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if constexpr (O == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if constexpr (O == 4) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else if constexpr (O == 8) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);  // row_mirror
    else return __shfl_xor(v, O, kWave);
}
template <int O>
__device__ __forceinline__ float partner(float v) { return __int_as_float(partner<O>(__float_as_int(v))); }

template <int G, int O = G / 2>
__device__ __forceinline__ int group_and(int v)
{
    if constexpr (O > 0) {
        // smallest distance first, so that the mirror steps pair fully reduced halves
        v = group_and<G, O / 2>(v);
        v &= partner<O>(v);
    }
    return v;
}

template <int G, int O = G / 2>
__device__ __forceinline__ void group_argmax(float& best_v, int& best_n, int& best_prop)
{
    if constexpr (O > 0) {
        group_argmax<G, O / 2>(best_v, best_n, best_prop);
        const float ov = partner<O>(best_v);
        const int on = partner<O>(best_n);
        const int op = partner<O>(best_prop);
        if (ov > best_v || (ov == best_v && on < best_n)) { best_v = ov; best_n = on; best_prop = op; }
    }
}

// Per-step, wave-uniform data of one update
struct PcStep {
    StepScalars sc;
    const float *q, *qbar, *qbar_tm1;
    int one, last_predictor_step;
    uint32_t draw, k0, k1, call8;
    FixedSoftmaxC2 fixed_c2;     // valid = 0 unless the caller evaluated it (persistent sampler, one atom type)
};

// Pointers of ONE structure (already offset to it); item0 = global index of its atom 0 in the Philox stream.
struct PcView {
    const int64_t* a;
    const float *x, *l, *logits, *score_x, *score_l, *z_coord, *gumbel, *u, *z_lat;
    const float* p2_table = nullptr;   // one atom type: this step's posterior in closed form (kP2Table floats), see below
    int64_t* a_out;
    float *x_out, *l_out, *p_out;
    int64_t item0, b;
};

__device__ __forceinline__ PcStep make_step(const PcArgs& p, int mode, int index, uint32_t draw_offset)
{
    PcStep st;
    const int C = p.C;
    st.q = p.q_explicit; st.qbar = p.qbar_explicit; st.qbar_tm1 = p.qbar_tm1_explicit;
    st.one = p.one_transition;
    st.last_predictor_step = 0;
    st.draw = draw_offset;
    if (p.use_tables) {
        st.sc = step_scalars(p.sched, mode, index, p.atoms_pow);
        st.q = p.sched.q + (int64_t)st.sc.idx * C * C;
        st.qbar = p.sched.qbar + (int64_t)st.sc.idx * C * C;
        st.qbar_tm1 = p.sched.qbar_tm1 + (int64_t)st.sc.idx * C * C;
        st.last_predictor_step = (mode == MDX_PREDICTOR && st.sc.idx == 0);
        if (st.last_predictor_step) st.one = 0;                 // generators/langevin_generator.py:601-604
        st.draw = (uint32_t)index * p.rng.draw_stride + draw_offset;
    }
    st.k0 = (uint32_t)p.rng.seed;
    st.k1 = (uint32_t)(p.rng.seed >> 32);
    st.call8 = rng_call(p.rng) << 8;
    st.fixed_c2 = FixedSoftmaxC2{0.0f, 0.0f, 0};
    return st;
}

__device__ __forceinline__ PcStep make_step(const PcArgs& p)
{
    return make_step(p, p.mode, (p.d_index ? *p.d_index : 0) + p.index_i, p.rng.draw_offset);
}

// The persistent sampler requests a step's schedule entries ONE SUB-STEP AHEAD, as VECTOR loads (every lane the same four
// addresses: one transaction each): a scalar load would share the LDS counter, and the first LDS wait of the forward would
// expose its latency; the vector counter is only waited on at the top of the next sub-step.  step_from_request() then builds
// the very StepScalars of step_scalars() from them (same loaded values, same operations).  (Four loads, not one load of a
// per-lane selected pointer: selecting among the addresses of fields of the argument structure makes the compiler keep the
// whole structure in scratch memory.)
struct StepRequest {
    float time, sigma, third, g;
};

__device__ __forceinline__ StepRequest step_request(const SchedDev& s, int mode, int index, int vzero)
{
    const bool corrector_at_zero = mode != MDX_PREDICTOR && index == 0;
    const int idx = (corrector_at_zero ? 0 : index - 1) + vzero;          // (vzero: a zero held in a vector register, so that
    StepRequest q;                                                        //  these are vector memory loads, scalar base + lane offset)
    q.time = s.time[idx];
    q.sigma = s.sigma[idx];
    q.third = mode == MDX_PREDICTOR ? s.g2[idx] : s.eps[index + vzero];
    q.g = s.g[idx];
    return q;
}

__device__ __forceinline__ StepScalars step_from_request(const SchedDev& s, int mode, int index, double atoms_pow, const StepRequest& q)
{
    StepScalars o;
    o.index = index;
    auto uniform = [](float v) {       // the value of the first active lane, as a scalar (bit pattern through the int builtin)
        return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
    };
    const float time = uniform(q.time), sigma = uniform(q.sigma), third = uniform(q.third), g = uniform(q.g);
    if (mode == MDX_PREDICTOR) {
        o.idx = index - 1;
        o.time = time;
        o.sigma = sigma;
        o.w = third;
        o.n = g;
        o.sigma_n = o.sigma / (float)atoms_pow;
    } else {
        if (index == 0) {
            o.idx = 0;
            o.time = 0.0f;
            o.sigma = (float)s.sigma_min;
            o.sigma_n = (float)(s.sigma_min / atoms_pow);
        } else {
            o.idx = index - 1;
            o.time = time;
            o.sigma = sigma;
            o.sigma_n = o.sigma / (float)atoms_pow;
        }
        o.w = third;
        o.n = __builtin_sqrtf(2.0f * o.w);
    }
    return o;
}

// make_step() of the persistent sampler (schedule tables always in use there): the scalars come from the request made a
// sub-step earlier, the Philox call word is read once per launch
__device__ __forceinline__ PcStep make_step_from_request(const PcArgs& p, int mode, int index, uint32_t draw_offset,
                                                         const StepRequest& q, uint32_t call8)
{
    PcStep st;
    const int C = p.C;
    st.sc = step_from_request(p.sched, mode, index, p.atoms_pow, q);
    st.q = p.sched.q + (int64_t)st.sc.idx * C * C;
    st.qbar = p.sched.qbar + (int64_t)st.sc.idx * C * C;
    st.qbar_tm1 = p.sched.qbar_tm1 + (int64_t)st.sc.idx * C * C;
    st.one = p.one_transition;
    st.last_predictor_step = (mode == MDX_PREDICTOR && st.sc.idx == 0);
    if (st.last_predictor_step) st.one = 0;                 // generators/langevin_generator.py:601-604
    st.draw = (uint32_t)index * p.rng.draw_stride + draw_offset;
    st.k0 = (uint32_t)p.rng.seed;
    st.k1 = (uint32_t)(p.rng.seed >> 32);
    st.call8 = call8;
    st.fixed_c2 = FixedSoftmaxC2{0.0f, 0.0f, 0};
    return st;
}

// The update of one structure by the G lanes of its group (P2, P1, P3).  Used by pc_step_kernel on global memory and
// by the fused MLP sampler kernel on its LDS-resident state: one body, one arithmetic.
template <int G>
__device__ __forceinline__ void pc_update_structure(const PcArgs& p, const PcStep& st, const PcView& v, int lane,
                                                    int update_types)
{
    const int N = p.N, C = p.C, d = p.d, M = p.C - 1;
    const StepScalars& sc = st.sc;
    const int one = st.one;
    int all_masked = 1;
    if (update_types && p.greedy) {
        for (int n = lane; n < N; n += G) all_masked &= (v.a[n] == M);
        all_masked = group_and<G>(all_masked);
    }
    float best_v = -__builtin_huge_valf();
    int best_n = 0x7fffffff;
    int best_prop = 0;
    for (int n = lane; n < N; n += G) {
        const uint32_t item = (uint32_t)(v.item0 + n);
        if (update_types) {
            const int a_t = (int)v.a[n];
            float pr[MDX_MAX_CLASSES], gm[MDX_MAX_CLASSES], lv[MDX_MAX_CLASSES];
            // One atom type, logits (finite, -inf): the posterior depends on the step and on a_t only, and the pre-pass
            // has evaluated it for both values of a_t (p2_table: p[MASK] and log(p[c] + eps)); otherwise evaluate it here.
            bool tabulated = false;
            float pr_mask = 0.0f, log_eps = 0.0f;
            if (C == 2 && v.p2_table && !v.p_out) {
                const float l0 = v.logits[n * C], l1 = v.logits[n * C + 1];
                if (l1 == -__builtin_huge_valf() && __builtin_fabsf(l0) < __builtin_huge_valf()) {
                    tabulated = true;
                    const int sel = a_t != 0;
                    pr_mask = v.p2_table[sel];
                    MDX_FOR_CLASSES(c) lv[c] = v.p2_table[2 + 2 * sel + c];
                    log_eps = v.p2_table[6];
                }
            }
            if (!tabulated) {
                posterior(v.logits + n * C, a_t, st.q, st.qbar, st.qbar_tm1, C, p.small_eps, pr, st.fixed_c2);
                MDX_FOR_CLASSES(c) if (c == M) pr_mask = pr[c];
            }
            if (v.gumbel) {
                MDX_FOR_CLASSES(c) gm[c] = v.gumbel[n * C + c];
            } else {
#pragma unroll
                for (int sub = 0; sub < MDX_MAX_CLASSES / 4; ++sub)
                    if (sub * 4 < C) {
                        const u32x4 r = philox4x32_10(item, st.call8 | (uint32_t)sub, st.draw, MDX_TAG_GUMBEL, st.k0, st.k1);
#pragma unroll
                        for (int l = 0; l < 4; ++l)
                            if (sub * 4 + l < C) gm[sub * 4 + l] = gumbel_from_u(u01(r.v[l]));
                    }
            }
            bool zero_mask = false;
            if (p.greedy) {                                      // :382-439
                float uu;
                if (v.u) uu = v.u[n];
                else uu = u01(philox4x32_10(item, st.call8, st.draw, MDX_TAG_BINARY, st.k0, st.k1).v[0]);
                const int unmask = uu > pr_mask;
                zero_mask = !all_masked && unmask && a_t == M;
                MDX_FOR_CLASSES(c) {
                    if (zero_mask && c == M) pr[c] = 0.0f;
                    if (!all_masked) gm[c] = 0.0f;
                }
            }
            if (tabulated) {
                MDX_FOR_CLASSES(c) if (zero_mask && c == M) lv[c] = log_eps;      // log(0 + eps)
            } else {
                MDX_FOR_CLASSES(c) lv[c] = logf_(pr[c] + p.small_eps);
            }
            float v_best = 0.0f;
            int prop = 0;
            MDX_FOR_CLASSES(c) {                                 // :311-315, first maximal index
                const float val = lv[c] + gm[c];
                if (c == 0 || val > v_best) { v_best = val; prop = c; }
                if (v.p_out) v.p_out[n * C + c] = pr[c];
            }
            if (one) {                                           // :339-380
                const float cand = (prop != a_t) ? v_best : -__builtin_huge_valf();
                if (cand > best_v || (cand == best_v && n < best_n)) { best_v = cand; best_n = n; best_prop = prop; }
                v.a_out[n] = a_t;
            } else {
                v.a_out[n] = prop;
                if (st.last_predictor_step && prop == M && p.status) atomicOr(p.status, MDX_STATUS_MASK_AT_LAST_STEP);
            }
        } else if (v.a_out && v.a_out != v.a) {
            v.a_out[n] = v.a[n];
        }
        if (p.do_coords) {                                       // :194-201
            float z0 = 0.0f, z1 = 0.0f, z2 = 0.0f, z3 = 0.0f;
            if (!v.z_coord) {
                const u32x4 r = philox4x32_10(item, st.call8, st.draw, MDX_TAG_COORD, st.k0, st.k1);
                box_muller(r.v[0], r.v[1], z0, z1);
                if (d > 2) box_muller(r.v[2], r.v[3], z2, z3);
            }
            // all loads first, then the three stores back to back (adjacent addresses -> one 12-byte store when d = 3)
            float xin[3], sin_[3], out[3];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < d) {
                    const int e = n * d + k;
                    const float zk = k == 0 ? z0 : (k == 1 ? z1 : z2);
                    xin[k] = v.x[e];
                    sin_[k] = v.score_x[e];
                    out[k] = v.z_coord ? v.z_coord[e] : zk;
                }
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < d) out[k] = coord_update(xin[k], sin_[k], out[k], sc.w, sc.n, sc.sigma);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < d) v.x_out[n * d + k] = out[k];
        }
    }
    if (update_types && one) {
        // arg-max over the atoms of the structure: larger value wins, ties go to the smaller atom index (a total
        // order, so the exchange pattern does not matter)
        group_argmax<G>(best_v, best_n, best_prop);
        if (lane == 0 && best_n < N) v.a_out[best_n] = best_prop;
    }
    if (p.do_lattice) {                                          // :475-490
        for (int k = lane; k < p.nl; k += G) {
            if (p.fixed_lattice) {
                if (v.l_out != v.l) v.l_out[k] = v.l[k];
            } else {
                float zz;
                if (v.z_lat) zz = v.z_lat[k];
                else {
                    const u32x4 r = philox4x32_10((uint32_t)v.b, st.call8 | (uint32_t)(k >> 2), st.draw, MDX_TAG_LATTICE,
                                                  st.k0, st.k1);
                    float z0, z1, z2, z3;
                    box_muller(r.v[0], r.v[1], z0, z1);
                    box_muller(r.v[2], r.v[3], z2, z3);
                    const int kk = k & 3;
                    zz = kk == 0 ? z0 : (kk == 1 ? z1 : (kk == 2 ? z2 : z3));
                }
                v.l_out[k] = (v.l[k] + (sc.w * v.score_l[k]) / sc.sigma_n) + sc.n * zz;
            }
        }
    }
}

__device__ __forceinline__ const float* off(const float* p, int64_t o) { return p ? p + o : nullptr; }
__device__ __forceinline__ float* off(float* p, int64_t o) { return p ? p + o : nullptr; }

// G lanes cooperate on one structure; a 64-lane wavefront carries 64/G structures.
// CSPEC > 0: number of classes (and d = 3) substituted as literals -- the class loops unroll to exactly C bodies and
// the per-atom index arithmetic folds; CSPEC = 0 is the generic instantiation.  Same code, same arithmetic.
template <int G, int CSPEC>
__global__ __launch_bounds__(kBlock) void pc_step_kernel(PcArgs p_in)
{
    PcArgs p = p_in;
    if constexpr (CSPEC > 0) { p.C = CSPEC; p.d = 3; p.nl = 6; }
    const int lane = threadIdx.x & (G - 1);
    const int64_t groups_per_grid = ((int64_t)gridDim.x * blockDim.x) / G;
    const int64_t group0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const int N = p.N, C = p.C, d = p.d;
    const PcStep st = make_step(p);
    for (int64_t b = group0; b < p.B; b += groups_per_grid) {
        PcView v;
        const int64_t a0 = b * N;
        v.a = p.a ? p.a + a0 : nullptr;
        v.x = off(p.x, a0 * d); v.l = off(p.l, b * p.nl);
        v.logits = off(p.logits, a0 * C); v.score_x = off(p.score_x, a0 * d); v.score_l = off(p.score_l, b * p.nl);
        v.z_coord = off(p.z_coord, a0 * d); v.gumbel = off(p.gumbel, a0 * C); v.u = off(p.u, a0);
        v.z_lat = off(p.z_lat, b * p.nl);
        v.a_out = p.a_out ? p.a_out + a0 : nullptr;
        v.x_out = off(p.x_out, a0 * d); v.l_out = off(p.l_out, b * p.nl); v.p_out = off(p.p_out, a0 * C);
        v.item0 = a0;
        v.b = b;
        pc_update_structure<G>(p, st, v, lane, p.update_types);
    }
}

int launch_pc(const PcArgs& a, hipStream_t st)
{
    // lanes per structure: smallest power of two >= N, capped at the wavefront
    int G = 1;
    while (G < a.N && G < kWave) G <<= 1;
    const int64_t threads = a.B * G;
    int64_t blocks = cdiv(threads, kBlock);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(kBlock);
    const int cspec = (a.d == 3 && (a.C == 2 || a.C == 3)) ? a.C : 0;
#define MDX_LAUNCH_PC(GG)                                                                                   \
    if (cspec == 2) hipLaunchKernelGGL((pc_step_kernel<GG, 2>), grid, block, 0, st, a);                       \
    else if (cspec == 3) hipLaunchKernelGGL((pc_step_kernel<GG, 3>), grid, block, 0, st, a);                  \
    else hipLaunchKernelGGL((pc_step_kernel<GG, 0>), grid, block, 0, st, a);
    switch (G) {
        case 1: MDX_LAUNCH_PC(1) break;
        case 2: MDX_LAUNCH_PC(2) break;
        case 4: MDX_LAUNCH_PC(4) break;
        case 8: MDX_LAUNCH_PC(8) break;
        case 16: MDX_LAUNCH_PC(16) break;
        case 32: MDX_LAUNCH_PC(32) break;
        default: MDX_LAUNCH_PC(64) break;
    }
#undef MDX_LAUNCH_PC
    return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Fused MLP score network + persistent sampler (one wavefront = one structure)
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMlpWaves = 4;                    // wavefronts (= structures) per workgroup, sharing one weight copy

// Explicit LDS pointers: with generic pointers hipcc emits flat_load, whose waits (vmcnt(0) AND lgkmcnt(0)) serialise
// the layer loops; address_space(3) pointers give ds_read_b128 with counted lgkmcnt waits.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) const float lds_cf;
typedef __attribute__((address_space(3))) int64_t lds_i64;
typedef __attribute__((address_space(3))) const int64_t lds_ci64;
typedef float lds_f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const lds_f4 lds_cf4;

// LDS exchanges between the lanes of ONE wavefront need no s_barrier (a wave's LDS operations execute in issue
// order); this only stops the compiler from moving memory operations across the hand-off.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// SiLU of the fused network forward: hardware exp2 / reciprocal (~1e-7 relative).  The forward is compared with the
// PyTorch module at 1e-5 (it is not part of the bit-exact MDX arithmetic contract, which covers the update kernels and
// the RNG); the IEEE sequence a / (1 + expf_(-a)) cost ~50 dependent instructions per layer on the critical path.
__device__ __forceinline__ float silu_(float a) { return __fdividef(a, 1.0f + __expf(-a)); }

// out[j] = bias[j] + sum_k in[k] * W[k][j]; lanes are output neurons, the input vector is broadcast from LDS four
// values at a time.  Four interleaved partial sums (k mod 4) shorten the dependent fmaf chain; they are added as
// ((s0+s1)+(s2+s3))+bias.  QUAD = true: the weights are the LDS image layout [k/4][j][k%4] (one 16-byte read per
// four k, rows zero-padded to a multiple of four); QUAD = false: plain transposed [k][j] in global memory.
template <bool QUAD, typename WP>
__device__ __forceinline__ void linear_wave(WP wt, WP bias, lds_cf* in, int in_dim, int out_dim, lds_f* out, int lane,
                                            bool silu_out)
{
    const int k4 = in_dim >> 2;
    lds_cf4* in4 = (lds_cf4*)in;
    for (int j = lane; j < out_dim; j += kWave) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if constexpr (QUAD) {
            lds_cf4* w = (lds_cf4*)wt + j;
            const int kq = (in_dim + 3) >> 2;
#pragma unroll 8
            for (int q = 0; q < kq; ++q) {
                const lds_f4 wv = w[q * out_dim];
                const lds_f4 v = in4[q];                                           // input padded with zeros
                s0 = __builtin_fmaf(wv.x, v.x, s0);
                s1 = __builtin_fmaf(wv.y, v.y, s1);
                s2 = __builtin_fmaf(wv.z, v.z, s2);
                s3 = __builtin_fmaf(wv.w, v.w, s3);
            }
        } else {
            WP w = wt + j;
#pragma unroll 2
            for (int q = 0; q < k4; ++q) {
                const lds_f4 v = in4[q];
                s0 = __builtin_fmaf(w[(4 * q + 0) * out_dim], v.x, s0);
                s1 = __builtin_fmaf(w[(4 * q + 1) * out_dim], v.y, s1);
                s2 = __builtin_fmaf(w[(4 * q + 2) * out_dim], v.z, s2);
                s3 = __builtin_fmaf(w[(4 * q + 3) * out_dim], v.w, s3);
            }
            for (int k = 4 * k4; k < in_dim; ++k) {
                const float t = __builtin_fmaf(w[k * out_dim], in[k], 0.0f);
                if ((k & 3) == 0) s0 += t; else if ((k & 3) == 1) s1 += t; else if ((k & 3) == 2) s2 += t; else s3 += t;
            }
        }
        float acc = ((s0 + s1) + (s2 + s3)) + bias[j];
        if (silu_out) acc = silu_(acc);
        out[j] = acc;
    }
}

// Offsets (in floats) of every parameter tensor inside one packed weight image, in the order they are staged.
struct MlpOffsets {
    int wc, bc, wn, bn, wt, bt, wa, ba, wl, bl, wh0, wh_first, wh_size, bh_size, woa, boa, total, in0;
    // hidden layer k: weights at wh(k), bias right behind them -- arithmetic instead of arrays (arrays indexed by a
    // loop variable end up in scratch memory)
    __host__ __device__ int wh(int k) const { return k == 0 ? wh0 : wh0 + wh_first + bh_size + (k - 1) * (wh_size + bh_size); }
    __host__ __device__ int bh(int k) const { return wh(k) + (k == 0 ? wh_first : wh_size); }
};

__host__ __device__ inline MlpOffsets mlp_offsets(const mdx_mlp_t& m)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2, H = m.hidden_size;
    MlpOffsets o;
    int t = 0;
    auto take = [&t](int n) { const int at = t; t += (n + 3) & ~3; return at; };     // 16-byte aligned pieces
    o.in0 = m.e_coordinates + m.e_noise + m.e_time + N * m.e_atom_type + m.e_lattice;
    auto quad = [](int k, int n) { return ((k + 3) & ~3) * n; };                     // [k/4][n][4] image of a [k][n] matrix
    o.wc = take(quad(2 * N * d, m.e_coordinates)); o.bc = take(m.e_coordinates);
    o.wn = take(m.e_noise); o.bn = take(m.e_noise);
    o.wt = take(m.e_time); o.bt = take(m.e_time);
    o.wa = take(C * m.e_atom_type); o.ba = take(m.e_atom_type);
    o.wl = take(nl * m.e_lattice); o.bl = take(m.e_lattice);
    o.wh_first = quad(o.in0, H);
    o.wh_size = quad(H, H);
    o.bh_size = (H + 3) & ~3;
    o.wh0 = take(o.wh_first + o.bh_size + (m.n_hidden - 1) * (o.wh_size + o.bh_size));
    // the three heads are staged as ONE [H][N C + N d + nl] matrix (one layer loop instead of three); woa/boa name it
    o.woa = take(quad(H, N * C + N * d + nl)); o.boa = take(N * C + N * d + nl);
    o.total = t;
    return o;
}

// The network's parameters as plain pointers (either the caller's global tensors or the LDS image)
struct MlpWeights {          // the caller's tensors in global memory
    const float *wc, *bc, *wn, *bn, *wt, *bt, *wa, *ba, *wl, *bl, *woa, *boa, *wox, *box, *wol, *bol;
    const mdx_mlp_t* m;
    __device__ const float* wh(int k) const { return m->w_hidden_t[k]; }
    __device__ const float* bh(int k) const { return m->b_hidden[k]; }
};
struct MlpWeightsLds {       // the workgroup's LDS image
    lds_cf *wc, *bc, *wn, *bn, *wt, *bt, *wa, *ba, *wl, *bl, *woa, *boa, *wox, *box, *wol, *bol, *img;
    MlpOffsets off;
    __device__ lds_cf* wh(int k) const { return img + off.wh(k); }
    __device__ lds_cf* bh(int k) const { return img + off.bh(k); }
};

__device__ __forceinline__ MlpWeights weights_global(const mdx_mlp_t& m)
{
    MlpWeights w;
    w.wc = m.w_coordinates_t; w.bc = m.b_coordinates; w.wn = m.w_noise_t; w.bn = m.b_noise;
    w.wt = m.w_time_t; w.bt = m.b_time; w.wa = m.w_atom_type_t; w.ba = m.b_atom_type;
    w.wl = m.w_lattice_t; w.bl = m.b_lattice;
    w.m = &m;
    w.woa = m.w_out_a_t; w.boa = m.b_out_a; w.wox = m.w_out_x_t; w.box = m.b_out_x; w.wol = m.w_out_l_t; w.bol = m.b_out_l;
    return w;
}

// Write every parameter tensor into an image (LDS or global) in the kernels' layout: matrices as [k/4][n][k%4] with
// zero rows up to a multiple of four, the three heads as one matrix.  Executed by all threads of one workgroup.
template <typename Dst>
__device__ __forceinline__ void write_mlp_image(const mdx_mlp_t& m, const MlpOffsets& o, Dst img)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2, H = m.hidden_size;
    auto copy = [&](const float* src, int at, int n) {
        for (int e = threadIdx.x; e < n; e += blockDim.x) img[at + e] = src[e];
    };
    auto copy_quad = [&](const float* src, int at, int k_dim, int n) {
        const int kp = (k_dim + 3) & ~3;
        for (int e = threadIdx.x; e < kp * n; e += blockDim.x) {
            const int q = e / (4 * n), r = e - q * 4 * n, j = r >> 2, c = r & 3, k = 4 * q + c;
            img[at + e] = k < k_dim ? src[k * n + j] : 0.0f;
        }
    };
    copy_quad(m.w_coordinates_t, o.wc, 2 * N * d, m.e_coordinates); copy(m.b_coordinates, o.bc, m.e_coordinates);
    copy(m.w_noise_t, o.wn, m.e_noise); copy(m.b_noise, o.bn, m.e_noise);
    copy(m.w_time_t, o.wt, m.e_time); copy(m.b_time, o.bt, m.e_time);
    copy(m.w_atom_type_t, o.wa, C * m.e_atom_type); copy(m.b_atom_type, o.ba, m.e_atom_type);
    copy(m.w_lattice_t, o.wl, nl * m.e_lattice); copy(m.b_lattice, o.bl, m.e_lattice);
    for (int k = 0; k < m.n_hidden; ++k) {
        copy_quad(m.w_hidden_t[k], o.wh(k), k == 0 ? o.in0 : H, H);
        copy(m.b_hidden[k], o.bh(k), H);
    }
    {   // merged heads: columns [0, NC) logits, [NC, NC+Nd) score_x, [NC+Nd, ..) score_l
        const int nt = N * C + N * d + nl, kp = (H + 3) & ~3;
        for (int e = threadIdx.x; e < kp * nt; e += blockDim.x) {
            const int q = e / (4 * nt), r = e - q * 4 * nt, j = r >> 2, c = r & 3, k = 4 * q + c;
            float val = 0.0f;
            if (k < H) {
                if (j < N * C) val = m.w_out_a_t[k * N * C + j];
                else if (j < N * C + N * d) val = m.w_out_x_t[k * N * d + (j - N * C)];
                else val = m.w_out_l_t[k * nl + (j - N * C - N * d)];
            }
            img[o.woa + e] = val;
        }
        for (int j = threadIdx.x; j < nt; j += blockDim.x)
            img[o.boa + j] = j < N * C ? m.b_out_a[j] : (j < N * C + N * d ? m.b_out_x[j - N * C] : m.b_out_l[j - N * C - N * d]);
    }
}

// Stage the weights into the workgroup's LDS image -- one coalesced 16-byte-per-lane copy when the caller supplied the
// packed image, the re-layout on the fly otherwise -- and point at the pieces.
__device__ __forceinline__ MlpWeightsLds weights_to_lds(const mdx_mlp_t& m, const MlpOffsets& o, lds_f* img)
{
    if (m.packed_image) {
        typedef __attribute__((address_space(3))) lds_f4 lds_wf4;
        const lds_f4* src = reinterpret_cast<const lds_f4*>(m.packed_image);
        lds_wf4* dst = (lds_wf4*)img;
        const int n4 = o.total >> 2;
#pragma unroll 8
        for (int e = threadIdx.x; e < n4; e += blockDim.x) dst[e] = src[e];     // 16-B loads in flight, 16-B LDS stores
    } else {
        write_mlp_image(m, o, img);
    }
    MlpWeightsLds w;
    lds_cf* c = (lds_cf*)img;
    w.wc = c + o.wc; w.bc = c + o.bc; w.wn = c + o.wn; w.bn = c + o.bn; w.wt = c + o.wt; w.bt = c + o.bt;
    w.wa = c + o.wa; w.ba = c + o.ba; w.wl = c + o.wl; w.bl = c + o.bl;
    w.woa = c + o.woa; w.boa = c + o.boa; w.wox = w.woa; w.box = w.boa; w.wol = w.woa; w.bol = w.boa;
    w.img = c;
    w.off = o;
    return w;
}

__global__ __launch_bounds__(kBlock) void mlp_pack_image_kernel(mdx_mlp_t m, float* image)
{
    const MlpOffsets o = mlp_offsets(m);
    for (int e = threadIdx.x; e < o.total; e += blockDim.x) image[e] = 0.0f;      // padding between the pieces
    __syncthreads();
    write_mlp_image(m, o, image);
}

// ---- generic instantiation with the input embeddings and the output heads FOLDED (any dimensions) ---------------------
// The same algebra as the register-resident family below (mdx_mlp_t.folded_input / folded_output, formed by the host in
// binary64 for any network): [cos | sin | sigma | t | atom-type embeddings | lattice embedding] -> hidden 0 is ONE linear map
// and (last hidden layer, three heads) is ONE linear map, so a forward is n_hidden layers instead of n_hidden + 2 and the
// software sincospi of the layer-by-layer form becomes the hardware v_cos / v_sin (argument in revolutions).  The two folded
// matrices ride behind the workgroup's LDS image.  Pays when the folded first layer is smaller than the two it replaces (small
// structures: the coordinate embedding 2 N d -> e_c is a low-rank factor for large N); the host decides (mlp_fold_pays).
__host__ __device__ inline int mlp_folded_inputs(const mdx_mlp_t& m)
{
    return 2 * m.number_of_atoms * m.spatial_dimension + 2 + m.number_of_atoms * m.e_atom_type + m.e_lattice;
}
__host__ __device__ inline int mlp_outputs(const mdx_mlp_t& m)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension;
    return N * m.num_classes + N * d + d * (d + 1) / 2;
}
// floats of the two folded blobs: [ceil(in / 4)][H][4] + bias [H]; [ceil(H / 4)][outputs][4] + bias [outputs]
__host__ __device__ inline int mlp_folded_in_floats(const mdx_mlp_t& m) { return ((mlp_folded_inputs(m) + 3) & ~3) * m.hidden_size + m.hidden_size; }
__host__ __device__ inline int mlp_folded_out_floats(const mdx_mlp_t& m) { return ((m.hidden_size + 3) & ~3) * mlp_outputs(m) + mlp_outputs(m); }
inline bool mlp_fold_pays(const mdx_mlp_t& m)
{
    if (!m.folded_input || !m.folded_output || m.n_hidden < 2) return false;
    const int64_t N = m.number_of_atoms, d = m.spatial_dimension, H = m.hidden_size;
    const int64_t in0 = m.e_coordinates + m.e_noise + m.e_time + N * m.e_atom_type + m.e_lattice;
    const int64_t plain = 2 * N * d * m.e_coordinates + in0 * H + H * H, folded = (int64_t)mlp_folded_inputs(m) * H;
    return folded < plain;      // (the heads' H x outputs product is common to both)
}

// MLPScoreNetwork forward for ONE structure by one wavefront (mlp_score_network.py:281-370).
// buf_a / buf_b: this wavefront's LDS scratch of mlp_scratch_floats() floats each.
template <bool QUAD, typename W>
__device__ __forceinline__ void mlp_forward_wave(const mdx_mlp_t& m, const W& w, int lane, lds_cf* x, lds_ci64* a, lds_cf* l,
                                                 float time, float sigma, lds_f* buf_a, lds_f* buf_b, lds_f* logits,
                                                 lds_f* score_x, lds_f* score_l)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2;
    const int nd = N * d;
    // (cos 2 pi x | sin 2 pi x), all cosines first (:299-305)
    for (int e = lane; e < nd; e += kWave) {
        float sn, cs;
        sincospif_(2.0f * x[e], sn, cs);
        buf_a[e] = cs;
        buf_a[nd + e] = sn;
    }
    if (lane < 4) buf_a[2 * nd + lane] = 0.0f;               // zero padding read by the four-wide layer loop
    wave_sync();
    // input of the first hidden layer: [coordinates | noise | time | atom types (N x e_atom_type) | lattice]
    int o = 0;
    linear_wave<QUAD>(w.wc, w.bc, buf_a, 2 * nd, m.e_coordinates, buf_b + o, lane, false);
    o += m.e_coordinates;
    for (int j = lane; j < m.e_noise; j += kWave) buf_b[o + j] = __builtin_fmaf(w.wn[j], sigma, w.bn[j]);
    o += m.e_noise;
    for (int j = lane; j < m.e_time; j += kWave) buf_b[o + j] = __builtin_fmaf(w.wt[j], time, w.bt[j]);
    o += m.e_time;
    for (int t = lane; t < N * m.e_atom_type; t += kWave) {      // Linear(one_hot(a)) = W[:, a] + b
        const int n = t / m.e_atom_type, e = t - n * m.e_atom_type;
        buf_b[o + t] = w.wa[(int)a[n] * m.e_atom_type + e] + w.ba[e];
    }
    o += N * m.e_atom_type;
    for (int j = lane; j < m.e_lattice; j += kWave) {
        float acc = w.bl[j];
        for (int k = 0; k < nl; ++k) acc = __builtin_fmaf(w.wl[k * m.e_lattice + j], l[k], acc);
        buf_b[o + j] = acc;
    }
    o += m.e_lattice;
    if (lane < 4) buf_b[o + lane] = 0.0f;
    wave_sync();
    // hidden stack: SiLU between layers, none after the last (:337-344)
    lds_f* in = buf_b;
    lds_f* out = buf_a;
    int in_dim = o;
    for (int k = 0; k < m.n_hidden; ++k) {
        linear_wave<QUAD>(w.wh(k), w.bh(k), in, in_dim, m.hidden_size, out, lane, k + 1 < m.n_hidden);
        if (lane < 4) out[m.hidden_size + lane] = 0.0f;
        wave_sync();
        lds_f* t = in; in = out; out = t;
        in_dim = m.hidden_size;
    }
    // heads; MASK logit forced to -inf (score_network.py:183-185)
    if constexpr (QUAD) {       // merged heads image; logits | score_x | score_l are contiguous in the wavefront's LDS
        linear_wave<QUAD>(w.woa, w.boa, in, in_dim, N * C + nd + nl, logits, lane, false);
    } else {
        linear_wave<QUAD>(w.woa, w.boa, in, in_dim, N * C, logits, lane, false);
        linear_wave<QUAD>(w.wox, w.box, in, in_dim, nd, score_x, lane, false);
        linear_wave<QUAD>(w.wol, w.bol, in, in_dim, nl, score_l, lane, false);
    }
    wave_sync();
    for (int n = lane; n < N; n += kWave) logits[n * C + C - 1] = -__builtin_huge_valf();
    wave_sync();
}

// The folded form of mlp_forward_wave (LDS image + the two folded blobs `fin`, `fout` in LDS); n_hidden >= 2.
__device__ __forceinline__ void mlp_forward_wave_folded(const mdx_mlp_t& m, const MlpWeightsLds& w, lds_cf* fin, lds_cf* fout,
                                                        int lane, lds_cf* x, lds_ci64* a, lds_cf* l, float time, float sigma,
                                                        lds_f* buf_a, lds_f* buf_b, lds_f* logits)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2, H = m.hidden_size;
    const int nd = N * d, ea = m.e_atom_type, el = m.e_lattice;
    const int in_f = mlp_folded_inputs(m), nt = mlp_outputs(m);
    for (int e = lane; e < nd; e += kWave) {
        const float xv = x[e];                                   // hardware cos / sin of 2 pi x: the argument is in revolutions
        buf_a[e] = __builtin_amdgcn_cosf(xv);
        buf_a[nd + e] = __builtin_amdgcn_sinf(xv);
    }
    if (lane == 0) buf_a[2 * nd] = sigma;
    if (lane == 1) buf_a[2 * nd + 1] = time;
    for (int t = lane; t < N * ea; t += kWave) {                 // Linear(one_hot(a)) = W[:, a] + b
        const int n = t / ea, e = t - n * ea;
        buf_a[2 * nd + 2 + t] = w.wa[(int)a[n] * ea + e] + w.ba[e];
    }
    for (int j = lane; j < el; j += kWave) {
        float acc = w.bl[j];
        for (int k = 0; k < nl; ++k) acc = __builtin_fmaf(w.wl[k * el + j], l[k], acc);
        buf_a[2 * nd + 2 + N * ea + j] = acc;
    }
    if (lane < 4) buf_a[in_f + lane] = 0.0f;                     // zero padding read by the four-wide layer loop
    wave_sync();
    linear_wave<true>(fin, fin + ((in_f + 3) & ~3) * H, buf_a, in_f, H, buf_b, lane, true);
    if (lane < 4) buf_b[H + lane] = 0.0f;
    wave_sync();
    lds_f* in = buf_b;
    lds_f* out = buf_a;
    for (int k = 1; k + 1 < m.n_hidden; ++k) {
        linear_wave<true>(w.wh(k), w.bh(k), in, H, H, out, lane, true);
        if (lane < 4) out[H + lane] = 0.0f;
        wave_sync();
        lds_f* t = in; in = out; out = t;
    }
    linear_wave<true>(fout, fout + ((H + 3) & ~3) * nt, in, H, nt, logits, lane, false);    // logits | score_x | score_l
    wave_sync();
    for (int n = lane; n < N; n += kWave) logits[n * C + C - 1] = -__builtin_huge_valf();   // MASK logit (score_network.py:183-185)
    wave_sync();
}

// ---- template-MLP specialisation: every layer's weights live in the lane's registers for the whole trajectory ----
// (one wavefront per SIMD, so the full 512-register file is available; 79 float4 = 316 VGPRs per lane)
struct MlpRegs {
    lds_f4 wc[12], wh0[19], wh1[16], wh2[16], wo[16];      // [k/4] quads of the lane's neuron: 48, 73(+3), 64, 64, 64 inputs
    float bc, bh0, bh1, bh2, bo;
};

__device__ __forceinline__ void load_mlp_regs(MlpRegs& R, const MlpWeightsLds& w, int lane)
{
    const lds_f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < 12; ++q) R.wc[q] = lane < 32 ? ((lds_cf4*)w.wc)[q * 32 + lane] : zero;
#pragma unroll
    for (int q = 0; q < 19; ++q) R.wh0[q] = ((lds_cf4*)w.wh(0))[q * 64 + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) R.wh1[q] = ((lds_cf4*)w.wh(1))[q * 64 + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) R.wh2[q] = ((lds_cf4*)w.wh(2))[q * 64 + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) R.wo[q] = lane < 46 ? ((lds_cf4*)w.woa)[q * 46 + lane] : zero;
    R.bc = lane < 32 ? w.bc[lane] : 0.0f;
    R.bh0 = w.bh(0)[lane]; R.bh1 = w.bh(1)[lane]; R.bh2 = w.bh(2)[lane];
    R.bo = lane < 46 ? w.boa[lane] : 0.0f;
}

// same partial sums and the same final ((s0+s1)+(s2+s3))+bias as linear_wave<true>
template <int KQ>
__device__ __forceinline__ float dot_regs(const lds_f4 (&wq)[KQ], lds_cf* in, float bias)
{
    // written on register PAIRS: (s0,s1) += w.xy * v.xy and (s2,s3) += w.zw * v.zw are one v_pk_fma_f32 each on the
    // halves of the 16-byte LDS read as they stand (left to itself the vectoriser pairs (x,w) / (y,z) and spends
    // three v_mov per quad re-packing the inputs).  Same four fmaf chains, same final sum.
    typedef float f2 __attribute__((ext_vector_type(2)));
    lds_cf4* in4 = (lds_cf4*)in;
    f2 s01 = {0.0f, 0.0f}, s23 = {0.0f, 0.0f};
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
        const lds_f4 v = in4[q];
        s01 = __builtin_elementwise_fma(wq[q].xy, v.xy, s01);
        s23 = __builtin_elementwise_fma(wq[q].zw, v.zw, s23);
    }
    return ((s01.x + s01.y) + (s23.x + s23.y)) + bias;
}



// mlp_forward_wave for the template dimensions (N 8, d 3, C 2, embeddings 32/16/16/1/1, hidden 64 x 3)
__device__ __forceinline__ void mlp_forward_regs(const MlpWeightsLds& w, const MlpRegs& R, int lane, lds_cf* x, lds_ci64* a,
                                                 lds_cf* l, float time, float sigma, lds_f* buf_a, lds_f* buf_b,
                                                 lds_f* logits)
{
    if (lane < 24) {
        float sn, cs;
        sincospif_(2.0f * x[lane], sn, cs);
        buf_a[lane] = cs;
        buf_a[24 + lane] = sn;
    }
    wave_sync();
    if (lane < 32) buf_b[lane] = dot_regs<12>(R.wc, buf_a, R.bc);
    else if (lane < 48) buf_b[lane] = __builtin_fmaf(w.wn[lane - 32], sigma, w.bn[lane - 32]);
    else buf_b[lane] = __builtin_fmaf(w.wt[lane - 48], time, w.bt[lane - 48]);
    if (lane < 8) buf_b[64 + lane] = w.wa[(int)a[lane]] + w.ba[0];
    if (lane == 8) {
        float acc = w.bl[0];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = __builtin_fmaf(w.wl[k], l[k], acc);
        buf_b[72] = acc;
    }
    if (lane >= 9 && lane < 12) buf_b[64 + lane] = 0.0f;        // zero padding of the 73-wide input up to 76
    wave_sync();
    buf_a[lane] = silu_(dot_regs<19>(R.wh0, buf_b, R.bh0));
    wave_sync();
    buf_b[lane] = silu_(dot_regs<16>(R.wh1, buf_a, R.bh1));
    wave_sync();
    buf_a[lane] = dot_regs<16>(R.wh2, buf_b, R.bh2);
    wave_sync();
    if (lane < 46) {                                            // logits (16) | score_x (24) | score_l (6), contiguous
        const float o = dot_regs<16>(R.wo, buf_a, R.bo);
        logits[lane] = (lane < 16 && (lane & 1)) ? -__builtin_huge_valf() : o;
    }
    wave_sync();
}

__host__ __device__ inline int mlp_scratch_floats(const mdx_mlp_t& m)
{
    const int N = m.number_of_atoms;
    const int in0 = m.e_coordinates + m.e_noise + m.e_time + N * m.e_atom_type + m.e_lattice;
    int mx = (mlp_folded_inputs(m) + 63) & ~63;              // >= 2 N d: the input vector of the folded first layer, padded to
    if (in0 > mx) mx = in0;                                  // the 64-value blocks the padded family reads (and >= 64 neurons)
    if (m.hidden_size > mx) mx = m.hidden_size;
    return ((mx + 3) & ~3) + 4;                              // + zero padding for the four-wide layer loop
}

// ---- template MLP with the input embeddings FOLDED into the first hidden layer (SPEC = 2) ---------------------------
// The five embedding layers of MLPScoreNetwork feed the first hidden layer with no activation in between
// (mlp_score_network.py:299-344), so  W_h0 [W_c [cos;sin] + b_c | w_n sigma + b_n | w_t t + b_t | emb_a | emb_l] + b_h0
// is ONE linear map of the 59-vector  [cos (24) | sin (24) | sigma | t | atom-type embeddings (8) | lattice embedding]:
// the host folds the products once (binary64, mdx_mlp_t.folded_input), and the forward loses a whole layer -- 12 + 19
// weight quads become 15, one LDS hand-off and 64 weight registers less.  Same function; the rounding differs from the
// layer-by-layer evaluation in the last bits (the fused path is compared with the PyTorch module at 1e-5, not bitwise).
// The same holds at the other end: the last hidden layer has no activation (:337-344), so it and the three output heads
// are one 64 -> 46 linear map (mdx_mlp_t.folded_output).  The template network is then three layers -- 15 + 16 + 16
// weight quads = 188 registers, no AGPR traffic -- instead of five.
// The family of register-resident instantiations: N = 8, d = 3, hidden 64, atom-type / lattice embeddings of size 1 (the
// reference's template), C in {2, 3} classes and NH in {2, 3, 4} hidden layers.  SPEC = 100 + 10 C + NH.
constexpr bool spec_folded(int SPEC) { return SPEC >= 100 && SPEC < 200; }
constexpr int spec_classes(int SPEC) { return SPEC >= 100 && SPEC < 200 ? (SPEC - 100) / 10 : 2; }
constexpr int spec_hidden_layers(int SPEC) { return SPEC >= 100 ? (SPEC % 100) % 10 : 3; }
// The PADDED register-resident family (SPEC = 200 + 10 (FQ / 16) + NH): any MLP with hidden <= 64, N <= 8 atoms,
// N C + N d + d (d + 1) / 2 <= 64 outputs, a folded input vector of <= 192 values and NH in {2, 3, 4} hidden layers -- every
// MLP configuration of the reference (its templates and experiments use hidden 16 .. 64 on 2 or 8 atoms with embeddings of 1
// .. 64).  The host pads the folded matrices with zeros to FIXED sizes (hidden -> 64 neurons, first-layer input -> FQ = 16, 32
// or 48 quads: mdx_mlp_t.folded_padded), so the layer loops have literal trip counts and the weights of all layers live in
// the lane's registers (up to 96 quads = 384 of the 512 a lone wavefront per SIMD may use); the dimensions of the structure
// (N, d, C, embedding sizes) stay run-time values in the input assembly and the update.  A zero quad adds fma(0, 0, s) = s:
// the same bits as the generic folded forward on the unpadded matrices (tests compare the two bit for bit).
constexpr bool spec_padded(int SPEC) { return SPEC >= 200; }
constexpr int spec_first_quads(int SPEC) { return SPEC >= 200 ? 16 * ((SPEC - 200) / 10) : 16; }

template <int C, int NH>
struct MlpRegsFolded {
    static constexpr int MID = NH - 2;                  // hidden layers between the folded input and the folded output
    static constexpr int NOUT = 8 * C + 24 + 6;         // logits | score_x | score_l
    lds_f4 wf[15], wmid[MID > 0 ? MID : 1][16], wfo[16];
    float bf, bmid[MID > 0 ? MID : 1], bfo;
};

template <int C, int NH>
__device__ __forceinline__ void load_mlp_regs_folded(MlpRegsFolded<C, NH>& R, const mdx_mlp_t& m, const MlpWeightsLds& w,
                                                     int lane)
{
    constexpr int NOUT = MlpRegsFolded<C, NH>::NOUT;
    const lds_f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
    const lds_f4* folded = reinterpret_cast<const lds_f4*>(m.folded_input);       // [15][64] quads, then the bias [64]
#pragma unroll
    for (int q = 0; q < 15; ++q) R.wf[q] = folded[q * 64 + lane];
    R.bf = m.folded_input[15 * 64 * 4 + lane];
#pragma unroll
    for (int k = 0; k < NH - 2; ++k) {
#pragma unroll
        for (int q = 0; q < 16; ++q) R.wmid[k][q] = ((lds_cf4*)w.wh(1 + k))[q * 64 + lane];
        R.bmid[k] = w.bh(1 + k)[lane];
    }
    const lds_f4* out = reinterpret_cast<const lds_f4*>(m.folded_output);        // [16][NOUT] quads, then the bias [NOUT]
#pragma unroll
    for (int q = 0; q < 16; ++q) R.wfo[q] = lane < NOUT ? out[q * NOUT + lane] : zero;
    R.bfo = lane < NOUT ? m.folded_output[16 * NOUT * 4 + lane] : 0.0f;
}

template <int C, int NH>
__device__ __forceinline__ void mlp_forward_folded(const MlpWeightsLds& w, const MlpRegsFolded<C, NH>& R, int lane, lds_cf* x,
                                                   lds_ci64* a, lds_cf* l, float time, float sigma, lds_f* buf_a, lds_f* buf_b,
                                                   lds_f* logits)
{
    constexpr int NOUT = MlpRegsFolded<C, NH>::NOUT;
    // the 59 (+1 zero) inputs of the folded layer
    if (lane < 24) {
        // hardware sin / cos take their argument in revolutions: cos(2 pi x), sin(2 pi x) directly
        const float xv = x[lane];
        buf_b[lane] = __builtin_amdgcn_cosf(xv);
        buf_b[24 + lane] = __builtin_amdgcn_sinf(xv);
    } else if (lane == 48) {
        buf_b[48] = sigma;
    } else if (lane == 49) {
        buf_b[49] = time;
    } else if (lane >= 50 && lane < 58) {
        buf_b[lane] = w.wa[(int)a[lane - 50]] + w.ba[0];
    } else if (lane == 58) {
        float acc = w.bl[0];
#pragma unroll
        for (int k = 0; k < 6; ++k) acc = __builtin_fmaf(w.wl[k], l[k], acc);
        buf_b[58] = acc;
    } else if (lane == 59) {
        buf_b[59] = 0.0f;
    }
    wave_sync();
    buf_a[lane] = silu_(dot_regs<15>(R.wf, buf_b, R.bf));
    wave_sync();
    // the middle layers ping-pong between the two buffers; `cur` ends up holding the input of the folded output layer
    lds_f* cur = buf_a;
    lds_f* other = buf_b;
#pragma unroll
    for (int k = 0; k < NH - 2; ++k) {
        other[lane] = silu_(dot_regs<16>(R.wmid[k], cur, R.bmid[k]));
        wave_sync();
        lds_f* t = cur; cur = other; other = t;
    }
    if (lane < NOUT) {                                          // logits (8 C) | score_x (24) | score_l (6), contiguous
        const float o = dot_regs<16>(R.wfo, cur, R.bfo);
        logits[lane] = (lane < 8 * C && (lane % C) == C - 1) ? -__builtin_huge_valf() : o;
    }
    // no hand-off here: the caller stores the step's noise record next to these outputs and synchronises once
}

template <int FQ, int NH>
struct MlpRegsPadded {
    static constexpr int MID = NH - 2;
    lds_f4 wf[FQ], wmid[MID > 0 ? MID : 1][16], wfo[16];
    float bf, bmid[MID > 0 ? MID : 1], bfo;
};

// mdx_mlp_t.folded_padded: [FQ][64][4] + bias [64] | (NH - 2) x ([16][64][4] + bias [64]) | [16][64][4] + bias [64]
template <int FQ, int NH>
__device__ __forceinline__ void load_mlp_regs_padded(MlpRegsPadded<FQ, NH>& R, const mdx_mlp_t& m, int lane)
{
    const lds_f4* blob = reinterpret_cast<const lds_f4*>(m.folded_padded);
#pragma unroll
    for (int q = 0; q < FQ; ++q) R.wf[q] = blob[q * 64 + lane];
    R.bf = m.folded_padded[FQ * 256 + lane];
    const float* at = m.folded_padded + FQ * 256 + 64;
#pragma unroll
    for (int k = 0; k < NH - 2; ++k) {
#pragma unroll
        for (int q = 0; q < 16; ++q) R.wmid[k][q] = reinterpret_cast<const lds_f4*>(at)[q * 64 + lane];
        R.bmid[k] = at[16 * 256 + lane];
        at += 16 * 256 + 64;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) R.wfo[q] = reinterpret_cast<const lds_f4*>(at)[q * 64 + lane];
    R.bfo = at[16 * 256 + lane];
}

template <int FQ, int NH>
__device__ __forceinline__ void mlp_forward_padded(const mdx_mlp_t& m, const MlpWeightsLds& w, const MlpRegsPadded<FQ, NH>& R,
                                                   int lane, lds_cf* x, lds_ci64* a, lds_cf* l, float time, float sigma,
                                                   lds_f* buf_a, lds_f* buf_b, lds_f* logits)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2;
    const int nd = N * d, ea = m.e_atom_type, el = m.e_lattice;
    const int in_f = mlp_folded_inputs(m), nt = mlp_outputs(m);
    // the input vector of the folded first layer, as mlp_forward_wave_folded builds it; zeros up to the padded length
    for (int e = lane; e < nd; e += kWave) {
        const float xv = x[e];
        buf_b[e] = __builtin_amdgcn_cosf(xv);
        buf_b[nd + e] = __builtin_amdgcn_sinf(xv);
    }
    if (lane == 0) buf_b[2 * nd] = sigma;
    if (lane == 1) buf_b[2 * nd + 1] = time;
    for (int t = lane; t < N * ea; t += kWave) {
        const int n = t / ea, e = t - n * ea;
        buf_b[2 * nd + 2 + t] = w.wa[(int)a[n] * ea + e] + w.ba[e];
    }
    for (int j = lane; j < el; j += kWave) {
        float acc = w.bl[j];
        for (int k = 0; k < nl; ++k) acc = __builtin_fmaf(w.wl[k * el + j], l[k], acc);
        buf_b[2 * nd + 2 + N * ea + j] = acc;
    }
    for (int e = in_f + lane; e < 4 * FQ; e += kWave) buf_b[e] = 0.0f;
    wave_sync();
    buf_a[lane] = silu_(dot_regs<FQ>(R.wf, buf_b, R.bf));       // (neurons beyond hidden_size: zero weights and bias -> 0)
    wave_sync();
    lds_f* cur = buf_a;
    lds_f* other = buf_b;
#pragma unroll
    for (int k = 0; k < NH - 2; ++k) {
        other[lane] = silu_(dot_regs<16>(R.wmid[k], cur, R.bmid[k]));
        wave_sync();
        lds_f* t = cur; cur = other; other = t;
    }
    if (lane < nt) {                                            // logits (N C) | score_x (N d) | score_l (nl), contiguous
        const float o = dot_regs<16>(R.wfo, cur, R.bfo);
        logits[lane] = (lane < N * C && (lane % C) == C - 1) ? -__builtin_huge_valf() : o;
    }
    // (no hand-off here: the caller synchronises once, behind the step's noise record)
}

// floats of LDS one wavefront needs besides the shared weight image
__host__ __device__ inline int mlp_wave_floats(const mdx_mlp_t& m)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2;
    const int f = 2 * mlp_scratch_floats(m) + 2 * N + N * d + ((nl + 3) & ~3) + N * C + N * d + ((nl + 3) & ~3) +
                  N * (d + C + 1) + 8;           // 8 = kP2Table
    return (f + 3) & ~3;
}

// LDS per wavefront: [buf_a S][buf_b S][a: N int64][x: N d][l: nl pad 4][logits: N C][score_x: N d][score_l: nl pad 4]
//                    [noise record of the current step: z N d | gumbel N C | u N]
struct MlpWaveLds {
    lds_f *buf_a, *buf_b, *x, *l, *logits, *sx, *sl, *noise;
    lds_i64* a;
};

__device__ __forceinline__ MlpWaveLds carve_wave_lds(const mdx_mlp_t& m, lds_f* base)
{
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2;
    const int S = mlp_scratch_floats(m);
    MlpWaveLds r;
    r.buf_a = base;
    r.buf_b = base + S;
    r.a = (lds_i64*)(base + 2 * S);                 // 8-byte aligned: every piece is a multiple of 4 floats
    r.x = base + 2 * S + 2 * N;
    r.l = r.x + N * d;
    r.logits = r.l + ((nl + 3) & ~3);
    r.sx = r.logits + N * C;
    r.sl = r.sx + N * d;
    r.noise = r.sl + ((nl + 3) & ~3);
    return r;
}

template <bool LDS_WEIGHTS>
__global__ __launch_bounds__(kMlpWaves* kWave) void mlp_forward_kernel(mdx_mlp_t m, const int64_t* a, const float* x,
                                                                       const float* l, const float* time,
                                                                       const float* sigma, int64_t batch, float* logits,
                                                                       float* score_x, float* score_l)
{
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    lds_f* lds = (lds_f*)lds_raw;
    const int N = m.number_of_atoms, d = m.spatial_dimension, C = m.num_classes, nl = d * (d + 1) / 2;
    const MlpOffsets off = mlp_offsets(m);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    auto run = [&](const auto& w, lds_f* scratch) {
        const MlpWaveLds r = carve_wave_lds(m, scratch + wave * mlp_wave_floats(m));
        for (int64_t b = (int64_t)blockIdx.x * kMlpWaves + wave; b < batch; b += (int64_t)gridDim.x * kMlpWaves) {
            for (int e = lane; e < N; e += kWave) r.a[e] = a[b * N + e];
            for (int e = lane; e < N * d; e += kWave) r.x[e] = x[b * N * d + e];
            for (int e = lane; e < nl; e += kWave) r.l[e] = l[b * nl + e];
            wave_sync();
            mlp_forward_wave<LDS_WEIGHTS>(m, w, lane, r.x, r.a, r.l, time[b], sigma[b], r.buf_a, r.buf_b, r.logits, r.sx, r.sl);
            for (int e = lane; e < N * C; e += kWave) logits[b * N * C + e] = r.logits[e];
            for (int e = lane; e < N * d; e += kWave) score_x[b * N * d + e] = r.sx[e];
            for (int e = lane; e < nl; e += kWave) score_l[b * nl + e] = r.sl[e];
            wave_sync();
        }
    };
    if constexpr (LDS_WEIGHTS) {
        const MlpWeightsLds w = weights_to_lds(m, off, lds);
        __syncthreads();
        run(w, lds + off.total);
    } else {
        run(weights_global(m), lds);
    }
}

struct MlpSampleArgs {
    PcArgs pc;               // flags, schedule, rng, dims (pointers unused)
    mdx_mlp_t mlp;
    int M, types_in_corrector, start_index, n_iterations;
    int fold;                // generic instantiation: the folded forward (mlp_forward_wave_folded); the blobs sit behind the image
    int diag_skip;           // diagnostics / tests (MDX_DIAG_SKIP bits): 1 = no forward, 2 = no update, 8 = no hoisted softmax
    int64_t* a;
    float *x, *l;
    // pre-drawn noise (pc_noise_fill_kernel): records [iteration][structure][predictor rec0 | M x corrector rec1];
    // NULL: every wavefront evaluates the Philox specification itself, on the few lanes its structure's atoms occupy
    const float* noise;
    int rec0, rec1;
};

// ---- noise pre-pass of the persistent sampler -----------------------------------------------------------------
// The draws of a trajectory do not depend on its state, so they are generated ahead of the loop by a kernel that
// fills the chip (one lane per atom and step) instead of inside the persistent kernel, where a structure's N atoms
// occupy N of the wavefront's 64 lanes and the ~600-instruction Philox / Box-Muller / Gumbel sequence sits on the
// critical path of every step.  Same counters, same functions => the same bits as the in-kernel draws.
constexpr int kP2Table = 8;   // per type-update record: p[MASK | a_t = 0, 1], log(p[c] + eps | a_t = 0, 1), log(eps), 1
struct NoiseFillArgs {
    mdx_rng_t rng;
    SchedDev sched;
    float small_eps;
    int start_index, n_iterations, types_in_corrector, greedy;
    int64_t B;
    int N, d, C, rec0, rec1, M;
    float* out;
};

// Records are assembled in LDS -- one lane per atom draws, as the persistent kernel would -- and leave the workgroup as
// contiguous runs (16 bytes per lane when the record geometry allows it): a record is read back as one stream, and one
// lane scattering 4-byte scalars into it wrote 36.4 MB for 29.5 MB of payload (1.23x, round-1 PMC).
constexpr int kStageFloats = 5120;      // >= (256 / N) (N (d + C + 1) + 8) for every N <= 64, d <= 3, C <= 8
__global__ __launch_bounds__(kBlock) void pc_noise_fill_kernel(NoiseFillArgs p)
{
    __shared__ __attribute__((aligned(16))) float stage[kStageFloats];
    const int N = p.N, d = p.d, C = p.C;
    const int64_t n_records = p.B * (int64_t)p.n_iterations;      // (iteration, structure) pairs
    const int rec_total = p.rec0 + p.M * p.rec1;                  // predictor part | M corrector parts
    const uint32_t k0 = (uint32_t)p.rng.seed, k1 = (uint32_t)(p.rng.seed >> 32);
    const uint32_t call8 = rng_call(p.rng) << 8;
    // whole records per workgroup pass: as many as the lanes cover (one lane per atom) and the stage holds
    int R = kBlock / N;
    if (R > kStageFloats / rec_total) R = kStageFloats / rec_total;
    const int r_local = threadIdx.x / N, n = threadIdx.x - r_local * N;
    const bool vec4 = !(rec_total & 3) && !(reinterpret_cast<uintptr_t>(p.out) & 15);
    for (int64_t r0 = (int64_t)blockIdx.x * R; r0 < n_records; r0 += (int64_t)gridDim.x * R) {
        const int64_t record = r0 + r_local;                      // = it * B + b
        if (r_local < R && record < n_records) {
            const int64_t it = record / p.B;
            const int64_t b = record - it * p.B;
            const int64_t item = b * N + n;
            const int i = p.start_index - 1 - (int)it;
            for (int sub = 0; sub <= p.M; ++sub) {                // 0 predictor, 1 + m corrector m
                const bool types = sub == 0 || p.types_in_corrector;
                const uint32_t draw = (uint32_t)(sub == 0 ? i + 1 : i) * p.rng.draw_stride + (uint32_t)sub;
                float* rec = stage + r_local * rec_total + (sub == 0 ? 0 : p.rec0 + (sub - 1) * p.rec1);
                const u32x4 r = philox4x32_10((uint32_t)item, call8, draw, MDX_TAG_COORD, k0, k1);
                float z0, z1, z2 = 0.0f, z3;
                box_muller(r.v[0], r.v[1], z0, z1);
                if (d > 2) box_muller(r.v[2], r.v[3], z2, z3);
                rec[n * d] = z0;
                if (d > 1) rec[n * d + 1] = z1;
                if (d > 2) rec[n * d + 2] = z2;
                if (!types) continue;
                for (int s4 = 0; s4 * 4 < C; ++s4) {
                    const u32x4 g = philox4x32_10((uint32_t)item, call8 | (uint32_t)s4, draw, MDX_TAG_GUMBEL, k0, k1);
                    for (int l = 0; l < 4 && s4 * 4 + l < C; ++l) rec[N * d + n * C + s4 * 4 + l] = gumbel_from_u(u01(g.v[l]));
                }
                // (the slot exists in every type record; without greedy sampling nobody reads it)
                rec[N * d + N * C + n] = p.greedy ? u01(philox4x32_10((uint32_t)item, call8, draw, MDX_TAG_BINARY, k0, k1).v[0])
                                                  : 0.0f;
                if (n == 0) {
                    // One atom type: with logits (finite, -inf) the posterior p(a_{t-1} | a_t) of this step is a function of
                    // a_t alone -- evaluated here once per step for a_t = 0 and a_t = MASK with the update's own functions
                    // (same bits), so that the update only selects.  Other class counts: table marked invalid.
                    float* t = rec + N * (d + C + 1);
                    for (int k = 0; k < kP2Table; ++k) t[k] = 0.0f;
                    if (C == 2) {
                        const int idx = sub == 0 ? i : (i > 0 ? i - 1 : 0);         // the step's row of the Q tables
                        const float lgc[2] = {0.0f, -__builtin_huge_valf()};
                        const FixedSoftmaxC2 fixed = fixed_softmax_c2(p.small_eps);
                        for (int sel = 0; sel < 2; ++sel) {
                            float pr[MDX_MAX_CLASSES];
                            posterior(lgc, sel, p.sched.q + idx * 4, p.sched.qbar + idx * 4, p.sched.qbar_tm1 + idx * 4, 2,
                                      p.small_eps, pr, fixed);
                            t[sel] = pr[1];
                            t[2 + 2 * sel] = logf_(pr[0] + p.small_eps);
                            t[3 + 2 * sel] = logf_(pr[1] + p.small_eps);
                        }
                        t[6] = logf_(0.0f + p.small_eps);
                        t[7] = 1.0f;
                    }
                }
            }
        }
        __syncthreads();
        // the staged records are ONE contiguous run of the workspace
        const int64_t here = n_records - r0 < R ? n_records - r0 : R;
        float* dst = p.out + r0 * rec_total;
        if (vec4) {
            for (int q = threadIdx.x; q < here * (rec_total >> 2); q += kBlock)
                reinterpret_cast<float4*>(dst)[q] = reinterpret_cast<const float4*>(stage)[q];
        } else {
            for (int q = threadIdx.x; q < here * rec_total; q += kBlock) dst[q] = stage[q];
        }
        __syncthreads();
    }
}

// One wavefront per structure, kMlpWaves structures per workgroup sharing one LDS image of the network's weights.
// The composition, the activations and the network outputs of a structure stay in its wavefront's LDS region for
// the whole trajectory segment; HBM is touched at the two ends only.  G = lanes that cooperate in the update.
// SPEC = 1: the dimensions of the reference's MLP template (config_diffusion_mlp.yaml with N = 8, d = 3: BASELINE
// configs 1-2; one atom type) are substituted as literals, so that after inlining every layer loop has constant trip
// counts and every LDS offset folds to an immediate.  Same code, same arithmetic; the host selects it when the
// descriptor matches, and tests compare it bit for bit with the generic instantiation.
template <int SPEC>
__device__ __forceinline__ void specialise(mdx_mlp_t& m, PcArgs& pc)
{
    if constexpr (SPEC >= 1 && SPEC < 200) {
        m.number_of_atoms = 8; m.spatial_dimension = 3; m.num_classes = spec_classes(SPEC); m.hidden_size = 64;
        m.n_hidden = spec_hidden_layers(SPEC);
        m.e_coordinates = 32; m.e_noise = 16; m.e_time = 16; m.e_atom_type = 1; m.e_lattice = 1;
        pc.N = 8; pc.d = 3; pc.C = spec_classes(SPEC); pc.nl = 6;
    }
}

inline bool matches_template_mlp(const mdx_mlp_t& m)
{
    return m.number_of_atoms == 8 && m.spatial_dimension == 3 && m.num_classes == 2 && m.hidden_size == 64 &&
           m.n_hidden == 3 && m.e_coordinates == 32 && m.e_noise == 16 && m.e_time == 16 && m.e_atom_type == 1 &&
           m.e_lattice == 1;
}

// the register-resident folded family (SPEC >= 100): one or two atom types, two to four hidden layers
inline int folded_family_spec(const mdx_mlp_t& m)
{
    const bool shape = m.number_of_atoms == 8 && m.spatial_dimension == 3 && m.hidden_size == 64 && m.e_coordinates == 32 &&
                       m.e_noise == 16 && m.e_time == 16 && m.e_atom_type == 1 && m.e_lattice == 1;
    if (!shape || m.num_classes < 2 || m.num_classes > 3 || m.n_hidden < 2 || m.n_hidden > 4) return 0;
    if (!m.folded_input || !m.folded_output) return 0;
    return 100 + 10 * m.num_classes + m.n_hidden;
}

template <int G, bool LDS_WEIGHTS, int SPEC>
__global__ __launch_bounds__(kMlpWaves* kWave) void mlp_pc_sample_kernel(MlpSampleArgs p_in)
{
    extern __shared__ __attribute__((aligned(16))) float lds_raw[];
    lds_f* lds = (lds_f*)lds_raw;
    MlpSampleArgs p = p_in;
    specialise<SPEC>(p.mlp, p.pc);
    const mdx_mlp_t& m = p.mlp;
    const int N = m.number_of_atoms, d = m.spatial_dimension, nl = d * (d + 1) / 2;
    const MlpOffsets off = mlp_offsets(m);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    // generic instantiation, folded forward: the two folded blobs behind the LDS image, in front of the wavefronts' regions
    [[maybe_unused]] lds_cf* fold_in = nullptr;
    [[maybe_unused]] lds_cf* fold_out = nullptr;
    auto run = [&](const auto& w, lds_f* scratch) {
        const MlpWaveLds r = carve_wave_lds(m, scratch + wave * mlp_wave_floats(m));
        [[maybe_unused]] MlpRegs regs;
        [[maybe_unused]] MlpRegsFolded<spec_classes(SPEC), spec_hidden_layers(SPEC)> folded;
        [[maybe_unused]] MlpRegsPadded<spec_first_quads(SPEC), spec_padded(SPEC) ? spec_hidden_layers(SPEC) : 2> padded;
        if constexpr (spec_padded(SPEC) && LDS_WEIGHTS) {
            load_mlp_regs_padded(padded, m, lane);
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0): as for the folded family below
        }
        if constexpr (SPEC == 1 && LDS_WEIGHTS) load_mlp_regs(regs, w, lane);
        if constexpr (spec_folded(SPEC) && LDS_WEIGHTS) {
            load_mlp_regs_folded(folded, m, w, lane);
            // The folded weights come from global memory.  Complete those loads HERE: left pending into the loop, the
            // compiler must assume them outstanding at the loop head and guards the first FMAs of every iteration with
            // vmcnt waits -- which, the counter being in-order, also wait for the noise record requested just before,
            // i.e. expose its whole latency in every sub-step.
            __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0), expcnt / lgkmcnt untouched
        }
        for (int64_t b = (int64_t)blockIdx.x * kMlpWaves + wave; b < p.pc.B; b += (int64_t)gridDim.x * kMlpWaves) {
            for (int e = lane; e < N; e += kWave) r.a[e] = p.a[b * N + e];
            for (int e = lane; e < N * d; e += kWave) r.x[e] = p.x[b * N * d + e];
            for (int e = lane; e < nl; e += kWave) r.l[e] = p.l[b * nl + e];
            wave_sync();
            PcView v;     // the update body addresses the same LDS through generic pointers (a handful of accesses)
            v.a = (const int64_t*)r.a; v.x = (const float*)r.x; v.l = (const float*)r.l;
            v.logits = (const float*)r.logits; v.score_x = (const float*)r.sx; v.score_l = (const float*)r.sl;
            v.z_coord = nullptr; v.gumbel = nullptr; v.u = nullptr; v.z_lat = nullptr;
            v.a_out = (int64_t*)r.a; v.x_out = (float*)r.x; v.l_out = (float*)r.l; v.p_out = nullptr;
            v.item0 = b * N;
            v.b = b;
            // one atom type: the clipped softmax of (l0, -inf) is a constant of the launch
            const FixedSoftmaxC2 fixed_c2 = (p.pc.C == 2 && !(p.diag_skip & 8)) ? fixed_softmax_c2(p.pc.small_eps)
                                                                                : FixedSoftmaxC2{0.0f, 0.0f, 0};
            PcArgs pc_types_only = p.pc;                                   // P2 + P3 only (P1 done one lane per component)
            pc_types_only.do_coords = 0;
            constexpr int kPre = SPEC >= 1 && SPEC < 200 ? 1 : (spec_padded(SPEC) ? 2 : MDX_MAX_CLASSES + 5);   // 64-lane fetches covering N (d + C + 1) + 8 floats (padded family: N <= 8, C <= 8 -> <= 104)
            const int rec_total = p.rec0 + p.M * p.rec1;
            // A sub-step's schedule entries do not depend on the state: they are REQUESTED one sub-step ahead, as vector loads (the
            // vector counter is waited on behind the forward, where the noise record is due anyway), instead of as scalar loads at
            // the top of the sub-step whose latency the forward's first instruction waited for (0.25 us of a 2.3 us sub-step).
            // Issued BEFORE the record's load: the counter is in order, and the wait for the record then covers them.
            const uint32_t call8_launch = rng_call(p.pc.rng) << 8;
            [[maybe_unused]] float pre[kPre];
            // (a zero the compiler cannot see through, in a vector register: with it in the index the table loads are not scalar loads)
            int vzero;
            asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
            StepRequest req_next = {0.0f, 0.0f, 0.0f, 0.0f};
            if (p.n_iterations > 0) req_next = step_request(p.pc.sched, MDX_PREDICTOR, p.start_index, vzero);   // the first sub-step's
            for (int it = 0; it < p.n_iterations; ++it) {
                const int i = p.start_index - 1 - it;               // loop variable of the reference (:147)
                for (int sub = 0; sub <= p.M; ++sub) {
                    const int mode = sub == 0 ? MDX_PREDICTOR : MDX_CORRECTOR;
                    PcStep st = make_step_from_request(p.pc, mode, sub == 0 ? i + 1 : i, (uint32_t)sub, req_next, call8_launch);
                    st.fixed_c2 = fixed_c2;
                    const int types = sub == 0 ? 1 : p.types_in_corrector;
                    const int rec_len = types ? p.rec0 : N * d;
                    {                                                // the NEXT sub-step's schedule entries
                        const bool wraps = sub == p.M;
                        const int nit = wraps ? it + 1 : it, nsub = wraps ? 0 : sub + 1;
                        if (nit < p.n_iterations) {
                            const int ni = p.start_index - 1 - nit;
                            req_next = step_request(p.pc.sched, nsub == 0 ? MDX_PREDICTOR : MDX_CORRECTOR, nsub == 0 ? ni + 1 : ni, vzero);
                        }
                    }
                    // this step's pre-drawn noise: fetched now, needed after the forward (the load's latency is hidden
                    // behind it), handed to the update through LDS
                    if (p.noise) {
                        const float* rec = p.noise + ((int64_t)it * p.pc.B + b) * rec_total +
                                           (sub == 0 ? 0 : p.rec0 + (sub - 1) * p.rec1);
#pragma unroll
                        for (int j = 0; j < kPre; ++j)
                            if (j * kWave + lane < rec_len) pre[j] = rec[j * kWave + lane];
                    }
                    if (!(p.diag_skip & 1)) {
                        if constexpr (SPEC == 1 && LDS_WEIGHTS)
                            mlp_forward_regs(w, regs, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma, r.buf_a, r.buf_b, r.logits);
                        else if constexpr (spec_folded(SPEC) && LDS_WEIGHTS)
                            mlp_forward_folded(w, folded, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma, r.buf_a, r.buf_b,
                                               r.logits);
                        else if constexpr (spec_padded(SPEC) && LDS_WEIGHTS)
                            mlp_forward_padded(m, w, padded, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma, r.buf_a, r.buf_b,
                                               r.logits);
                        else if constexpr (SPEC == 0 && LDS_WEIGHTS) {
                            if (p.fold)
                                mlp_forward_wave_folded(m, w, fold_in, fold_out, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma,
                                                        r.buf_a, r.buf_b, r.logits);
                            else
                                mlp_forward_wave<LDS_WEIGHTS>(m, w, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma, r.buf_a,
                                                              r.buf_b, r.logits, r.sx, r.sl);
                        } else
                            mlp_forward_wave<LDS_WEIGHTS>(m, w, lane, r.x, r.a, r.l, st.sc.time, st.sc.sigma, r.buf_a,
                                                          r.buf_b, r.logits, r.sx, r.sl);
                    }
                    if (p.noise) {
#pragma unroll
                        for (int j = 0; j < kPre; ++j)
                            if (j * kWave + lane < rec_len) r.noise[j * kWave + lane] = pre[j];
                    }
                    wave_sync();                                     // network outputs + noise record visible to the update
                    if (p.noise) {
                        v.z_coord = (const float*)r.noise;
                        v.gumbel = (const float*)(r.noise + N * d);
                        v.u = p.pc.greedy ? (const float*)(r.noise + N * d + N * p.pc.C) : nullptr;
                        v.p2_table = (types && r.noise[N * (d + p.pc.C + 1) + 7] == 1.0f && !(p.diag_skip & 16))
                                         ? (const float*)(r.noise + N * (d + p.pc.C + 1)) : nullptr;
                    }
                    if (p.noise && !(p.diag_skip & 2)) {
                        // with pre-drawn noise the coordinate update is elementwise over the N d components: one lane
                        // per component (24 lanes at C2) instead of three components in sequence on each atom's lane
                        for (int e = lane; e < N * d; e += kWave)
                            r.x[e] = coord_update(r.x[e], r.sx[e], r.noise[e], st.sc.w, st.sc.n, st.sc.sigma);
                        if (lane < G) pc_update_structure<G>(pc_types_only, st, v, lane, types);
                    } else if (lane < G && !(p.diag_skip & 2)) {
                        pc_update_structure<G>(p.pc, st, v, lane, types);
                    }
                    wave_sync();
                }
            }
            for (int e = lane; e < N; e += kWave) p.a[b * N + e] = r.a[e];
            for (int e = lane; e < N * d; e += kWave) p.x[b * N * d + e] = r.x[e];
            for (int e = lane; e < nl; e += kWave) p.l[b * nl + e] = r.l[e];
            wave_sync();
        }
    };
    if constexpr (LDS_WEIGHTS) {
        const MlpWeightsLds w = weights_to_lds(m, off, lds);
        int extra = 0;
        if constexpr (SPEC == 0) {
            if (p.fold) {
                const int n_in = mlp_folded_in_floats(m), n_out = mlp_folded_out_floats(m);
                lds_f* blob = lds + off.total;
                for (int e = threadIdx.x; e < n_in; e += blockDim.x) blob[e] = m.folded_input[e];
                for (int e = threadIdx.x; e < n_out; e += blockDim.x) blob[n_in + e] = m.folded_output[e];
                fold_in = blob;
                fold_out = blob + n_in;
                extra = (n_in + n_out + 3) & ~3;
            }
        }
        __syncthreads();
        run(w, lds + off.total + extra);
    } else {
        run(weights_global(m), lds);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// F2 stand-alone and R1
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int noised_atom_type(int a0, const float* __restrict__ qbar, int C, const float* gum_u,
                                                bool from_u)
{
    float best = 0.0f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
        const float lq = logf_(qbar[a0 * C + c]);
        const float gn = from_u ? gumbel_from_u(gum_u[c]) : gum_u[c];
        const float v = lq + gn;
        if (c == 0 || v > best || (v != v && best == best)) { best = v; arg = c; }
    }
    return arg;
}

// atom_stride: 0 = one [C,C] matrix for every atom, C*C = a matrix per atom (the reference's broadcast q_bar [..., C, C])
__global__ __launch_bounds__(kBlock) void noise_atom_types_kernel(const int64_t* a0, const float* qbar, int64_t atom_stride,
                                                                  const float* u, int64_t n_atoms, int C, int64_t* out)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_atoms; i += (int64_t)gridDim.x * blockDim.x) {
        float uu[MDX_MAX_CLASSES];
        for (int c = 0; c < C; ++c) uu[c] = u[i * C + c];
        out[i] = noised_atom_type((int)a0[i], qbar + i * atom_stride, C, uu, true);
    }
}

struct RepaintArgs {
    SchedDev sched;
    int index_i;
    const int32_t* d_index;
    const float* cx;
    const int64_t *ca, *cidx;
    int K;
    const float *z, *u;
    mdx_rng_t rng;
    int64_t B;
    int N, d, C;
    float* x;
    int64_t* a;
};

__global__ __launch_bounds__(kBlock) void repaint_rows_kernel(RepaintArgs p)
{
    const int index = (p.d_index ? *p.d_index : 0) + p.index_i;
    const int C = p.C, d = p.d;
    const uint32_t k0 = (uint32_t)p.rng.seed, k1 = (uint32_t)(p.rng.seed >> 32);
    const uint32_t call8 = rng_call(p.rng) << 8;
    const uint32_t draw = (uint32_t)index * p.rng.draw_stride + p.rng.draw_offset;
    const int64_t total = p.B * p.K;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = t / p.K;
        const int k = (int)(t - b * p.K);
        const int64_t row = p.cidx[k];
        const int64_t at = b * p.N + row;
        const int a0 = (int)p.ca[k];
        if (index == 0) {                        // constrained_langevin_generator.py:120-123: no noise at t = 0
            for (int c = 0; c < d; ++c) p.x[at * d + c] = p.cx[k * d + c];
            p.a[at] = a0;
            continue;
        }
        const int idx = index - 1;               // noising_transform.py:112
        const float sigma = p.sched.sigma[idx];
        float z[4];
        if (!p.z) {
            const u32x4 r = philox4x32_10((uint32_t)at, call8, draw, MDX_TAG_REPAINT_Z, k0, k1);
            box_muller(r.v[0], r.v[1], z[0], z[1]);
            if (d > 2) box_muller(r.v[2], r.v[3], z[2], z[3]);
        }
        for (int c = 0; c < d; ++c) {
            const float zz = p.z ? p.z[at * d + c] : z[c];
            p.x[at * d + c] = wrap01(p.cx[k * d + c] + sigma * zz);
        }
        float uu[MDX_MAX_CLASSES];
        if (p.u) {
            for (int c = 0; c < C; ++c) uu[c] = p.u[at * C + c];
        } else {
            for (int sub = 0; sub * 4 < C; ++sub) {
                const u32x4 r = philox4x32_10((uint32_t)at, call8 | (uint32_t)sub, draw, MDX_TAG_REPAINT_U, k0, k1);
                for (int l = 0; l < 4 && sub * 4 + l < C; ++l) uu[sub * 4 + l] = u01(r.v[l]);
            }
        }
        p.a[at] = noised_atom_type(a0, p.sched.qbar + (int64_t)idx * C * C, C, uu, true);
    }
}

// RePaint resampling: one forward-process step i -> i+1 on every atom (no reference counterpart; include/mdx_hip.h)
struct ForwardStepArgs {
    SchedDev sched;
    int index_i;
    const int32_t* d_index;
    const float *z, *u;
    mdx_rng_t rng;
    int64_t atoms;
    int d, C;
    float* x;
    int64_t* a;
};

__global__ __launch_bounds__(kBlock) void forward_step_kernel(ForwardStepArgs p)
{
    const int index = (p.d_index ? *p.d_index : 0) + p.index_i;
    if (index < 1 || index >= p.sched.T) return;          // nothing to re-noise at the ends of the trajectory
    const int C = p.C, d = p.d;
    const uint32_t k0 = (uint32_t)p.rng.seed, k1 = (uint32_t)(p.rng.seed >> 32);
    const uint32_t call8 = rng_call(p.rng) << 8;
    const uint32_t draw = (uint32_t)index * p.rng.draw_stride + p.rng.draw_offset;
    const float g = p.sched.g[index];
    const float* q = p.sched.q + (int64_t)index * C * C;
    for (int64_t at = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; at < p.atoms; at += (int64_t)gridDim.x * blockDim.x) {
        float z[4];
        if (!p.z) {
            const u32x4 r = philox4x32_10((uint32_t)at, call8, draw, MDX_TAG_RESAMPLE_Z, k0, k1);
            box_muller(r.v[0], r.v[1], z[0], z[1]);
            if (d > 2) box_muller(r.v[2], r.v[3], z[2], z[3]);
        }
        for (int c = 0; c < d; ++c) {
            const float zz = p.z ? p.z[at * d + c] : z[c];
            p.x[at * d + c] = wrap01(p.x[at * d + c] + g * zz);
        }
        float uu[MDX_MAX_CLASSES];
        if (p.u) {
            for (int c = 0; c < C; ++c) uu[c] = p.u[at * C + c];
        } else {
            for (int sub = 0; sub * 4 < C; ++sub) {
                const u32x4 r = philox4x32_10((uint32_t)at, call8 | (uint32_t)sub, draw, MDX_TAG_RESAMPLE_U, k0, k1);
                for (int l = 0; l < 4 && sub * 4 + l < C; ++l) uu[sub * 4 + l] = u01(r.v[l]);
            }
        }
        p.a[at] = noised_atom_type((int)p.a[at], q, C, uu, true);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// N1: radius graph
// ---------------------------------------------------------------------------------------------------------------
constexpr int kRowsPerBlock = 16;

__device__ __forceinline__ float crossing_distance(const float* cell)
{
    const float* a1 = cell; const float* a2 = cell + 3; const float* a3 = cell + 6;
    const float c12x = a1[1] * a2[2] - a1[2] * a2[1], c12y = a1[2] * a2[0] - a1[0] * a2[2], c12z = a1[0] * a2[1] - a1[1] * a2[0];
    const float c13x = a1[1] * a3[2] - a1[2] * a3[1], c13y = a1[2] * a3[0] - a1[0] * a3[2], c13z = a1[0] * a3[1] - a1[1] * a3[0];
    const float c23x = a2[1] * a3[2] - a2[2] * a3[1], c23y = a2[2] * a3[0] - a2[0] * a3[2], c23z = a2[0] * a3[1] - a2[1] * a3[0];
    const float vol = __builtin_fabsf((c12x * a3[0] + c12y * a3[1]) + c12z * a3[2]);
    const float n12 = __builtin_sqrtf((c12x * c12x + c12y * c12y) + c12z * c12z);
    const float n13 = __builtin_sqrtf((c13x * c13x + c13y * c13y) + c13z * c13z);
    const float n23 = __builtin_sqrtf((c23x * c23x + c23y * c23y) + c23z * c23z);
    float dmin = vol / n12;
    const float d2 = vol / n13;
    if (d2 < dmin) dmin = d2;
    const float d3 = vol / n23;
    if (d3 < dmin) dmin = d3;
    return dmin;
}

// Bit l of the result: image l (itertools.product(-1, 0, 1) order) of atom j lies within the cutoff of atom i, 0 < d^2 <= rc^2
// (neighbors.py:192-194 excludes coincident atoms as well as the atom itself).  `ortho`: the cell is diagonal with
// rc <= L_min / 2.2, so only the nearest image can qualify and ONE is evaluated, with the expression of the sweep: same bits.
__device__ __forceinline__ uint32_t images_within_cutoff(bool ortho, float pix, float piy, float piz, float pjx, float pjy, float pjz,
                                                         float inv_lx, float inv_ly, float inv_lz, float lx, float ly, float lz,
                                                         const float* lv, float rc2)
{
    uint32_t mask = 0;
    if (ortho) {
        const int nx = max(-1, min(1, (int)__builtin_rintf((pix - pjx) * inv_lx)));
        const int ny = max(-1, min(1, (int)__builtin_rintf((piy - pjy) * inv_ly)));
        const int nz = max(-1, min(1, (int)__builtin_rintf((piz - pjz) * inv_lz)));
        // image vector of a diagonal cell: n_k * L_k, exact, identical to the fma chain that fills lv[]
        const float sx = pjx + (float)nx * lx, sy = pjy + (float)ny * ly, sz = pjz + (float)nz * lz;
        const float dx = pix - sx, dy = piy - sy, dz = piz - sz;
        const float d2 = (dx * dx + dy * dy) + dz * dz;
        if (0.0f < d2 && d2 <= rc2) mask = (1u << ((nx + 1) * 9 + (ny + 1) * 3 + (nz + 1)));
    } else {
        // not unrolled: a full unroll hoists the 81 image-vector components into registers (113 VGPRs, half the
        // occupancy) for the benefit of the rare triclinic path
#pragma nounroll
        for (int l = 0; l < 27; ++l) {
            const float sx = pjx + lv[3 * l], sy = pjy + lv[3 * l + 1], sz = pjz + lv[3 * l + 2];
            const float dx = pix - sx, dy = piy - sy, dz = piz - sz;
            const float d2 = (dx * dx + dy * dy) + dz * dz;
            if (0.0f < d2 && d2 <= rc2) mask |= (1u << l);
        }
    }
    return mask;
}

// images_within_cutoff(...) != 0 for a caller that does not need to know WHICH image: the image number stays a float (rint, then
// clamped to -1 .. 1 by a median) instead of going through an integer -- the same n, the same n * L, pj + n * L, pi - (...) and d^2,
// ten instructions fewer per pair.  (A NaN difference gives n = 0 there and an unspecified n here; d^2 is NaN either way: no hit.)
__device__ __forceinline__ bool any_image_within_cutoff(bool ortho, float pix, float piy, float piz, float pjx, float pjy, float pjz,
                                                        float inv_lx, float inv_ly, float inv_lz, float lx, float ly, float lz,
                                                        const float* lv, float rc2)
{
    if (!ortho) return images_within_cutoff(false, pix, piy, piz, pjx, pjy, pjz, inv_lx, inv_ly, inv_lz, lx, ly, lz, lv, rc2) != 0;
    const float nx = __builtin_amdgcn_fmed3f(__builtin_rintf((pix - pjx) * inv_lx), -1.0f, 1.0f);
    const float ny = __builtin_amdgcn_fmed3f(__builtin_rintf((piy - pjy) * inv_ly), -1.0f, 1.0f);
    const float nz = __builtin_amdgcn_fmed3f(__builtin_rintf((piz - pjz) * inv_lz), -1.0f, 1.0f);
    const float sx = pjx + nx * lx, sy = pjy + ny * ly, sz = pjz + nz * lz;
    const float dx = pix - sx, dy = piy - sy, dz = piz - sz;
    const float d2 = (dx * dx + dy * dy) + dz * dz;
    return 0.0f < d2 && d2 <= rc2;
}

// One workgroup = one (structure, chunk of kRowsPerBlock source rows).  The structure's positions and its 27
// image vectors are staged in LDS once; each wavefront then owns source rows and sweeps the destinations 64 at
// a time.  Lane ranks from ballot/scan make the writes dense and ordered by (src, dst, image).
template <bool FILL>
__global__ __launch_bounds__(kBlock) void radius_graph_kernel(const float* __restrict__ cart, const float* __restrict__ cell,
                                                              float rc, int64_t B, int N, int unique, int chunks,
                                                              int64_t* __restrict__ counts, const int64_t* __restrict__ offsets,
                                                              int64_t* __restrict__ edges, int32_t* __restrict__ image_out,
                                                              float* __restrict__ shifts_out, uint32_t* status,
                                                              int64_t capacity, const float* __restrict__ lattice,
                                                              int lattice_stride, float clip_min)
{
    extern __shared__ float lds[];
    float* pos = lds;            // [N][3]
    float* lv = lds + 3 * N;     // [27][3]
    const int64_t b = blockIdx.x / chunks;
    const int chunk = blockIdx.x % chunks;
    const float* P = cart + b * N * 3;
    // lattice != nullptr (the EGNN score network's graph, egnn_score_network.py:236-247): `cart` holds RELATIVE coordinates and
    // the cell is diag(max(lattice[b, k], clip_min)); relative x diagonal cell is one product per component -- the bits
    // torch.matmul(relative, diag_embed(lengths)) gives (its other terms are exact zeros)
    float cl[9];
    if (lattice) {
#pragma unroll
        for (int k = 0; k < 9; ++k) cl[k] = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = lattice[b * lattice_stride + k];
            cl[4 * k] = v < clip_min ? clip_min : v;          // torch.clip(min=): a NaN stays a NaN
        }
        for (int i = threadIdx.x; i < 3 * N; i += blockDim.x) {
            const int c = i % 3;
            pos[i] = P[i] * (c == 0 ? cl[0] : c == 1 ? cl[4] : cl[8]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) cl[k] = cell[b * 9 + k];
        for (int i = threadIdx.x; i < 3 * N; i += blockDim.x) pos[i] = P[i];
    }
    if (threadIdx.x < 81) {
        const int l = threadIdx.x / 3, c = threadIdx.x % 3;
        const float rel[3] = {(float)(l / 9 - 1), (float)((l / 3) % 3 - 1), (float)(l % 3 - 1)};
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = __builtin_fmaf(rel[k], c == 0 ? cl[k * 3] : c == 1 ? cl[k * 3 + 1] : cl[k * 3 + 2], acc);
        lv[threadIdx.x] = acc;
    }
    if (!FILL && chunk == 0 && threadIdx.x == 96 && status) {
        if (!(crossing_distance(cl) > rc)) atomicOr(status, MDX_STATUS_CUTOFF_TOO_LARGE);
    }
    // Orthorhombic cell with rc <= L_min / 2.2 (always true on the EGNN path, which clips the cell to 2.2 rc):
    // an image within rc has every component |delta_k| <= rc <= 0.4546 L_k, so it is THE nearest image and the
    // other 26 cannot qualify.  One image is then evaluated -- with the same expression, hence the same bits.
    const bool ortho = cl[1] == 0.0f && cl[2] == 0.0f && cl[3] == 0.0f && cl[5] == 0.0f && cl[6] == 0.0f &&
                       cl[7] == 0.0f && cl[0] > 0.0f && cl[4] > 0.0f && cl[8] > 0.0f &&
                       rc * 2.2f <= fminf(cl[0], fminf(cl[4], cl[8]));
    const float inv_lx = ortho ? 1.0f / cl[0] : 0.0f, inv_ly = ortho ? 1.0f / cl[4] : 0.0f,
                inv_lz = ortho ? 1.0f / cl[8] : 0.0f;
    __syncthreads();
    const float rc2 = rc * rc;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int row_end = min(N, (chunk + 1) * kRowsPerBlock);
    for (int i = chunk * kRowsPerBlock + wave; i < row_end; i += kBlock / kWave) {
        const float pix = pos[3 * i], piy = pos[3 * i + 1], piz = pos[3 * i + 2];
        const int64_t row = b * N + i;
        const int64_t base = FILL ? offsets[row] : 0;
        int64_t running = 0;
        for (int j0 = 0; j0 < N; j0 += kWave) {
            const int j = j0 + lane;
            uint32_t mask = 0;
            if (j < N)
                mask = images_within_cutoff(ortho, pix, piy, piz, pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], inv_lx, inv_ly,
                                            inv_lz, cl[0], cl[4], cl[8], lv, rc2);
            int cnt, rank, total;
            if (unique) {
                cnt = (mask != 0);
                const unsigned long long hits = __ballot(cnt);
                rank = __popcll(hits & ((1ull << lane) - 1ull));
                total = __popcll(hits);
            } else {
                cnt = __popc(mask);
                int incl = cnt;                      // inclusive prefix over the lanes
#pragma unroll
                for (int o = 1; o < kWave; o <<= 1) {
                    const int v = __shfl_up(incl, o, kWave);
                    if (lane >= o) incl += v;
                }
                total = __shfl(incl, kWave - 1, kWave);
                rank = incl - cnt;
            }
            if (FILL && cnt) {
                int64_t e = base + running + rank;
                if (unique) {
                    longlong2 pair;
                    pair.x = row;
                    pair.y = row - i + j;
                    if (e < capacity) reinterpret_cast<longlong2*>(edges)[e] = pair;      // one 16-B store per edge
                } else {
                    uint32_t m = mask;
                    while (m && e < capacity) {
                        const int l = __ffs(m) - 1;
                        m &= m - 1;
                        longlong2 pair;
                        pair.x = i;
                        pair.y = j;
                        reinterpret_cast<longlong2*>(edges)[e] = pair;
                        image_out[e] = l;
                        if (shifts_out) {
                            shifts_out[3 * e] = lv[3 * l];
                            shifts_out[3 * e + 1] = lv[3 * l + 1];
                            shifts_out[3 * e + 2] = lv[3 * l + 2];
                        }
                        ++e;
                    }
                }
            }
            running += total;
        }
        if (!FILL && lane == 0) counts[row] = running;
        // a caller-sized edge list that is too small: nothing is written beyond it, and the caller is told
        if (FILL && lane == 0 && status && base + running > capacity) atomicOr(status, MDX_STATUS_GRAPH_CAPACITY);
    }
}

// offsets[i] = counts[0] + ... + counts[i-1], *total = the sum: ONE workgroup walks the list in tiles of 16 384 entries (the list is
// the per-atom edge count of a batch -- 32 768 entries, two tiles, at C3; a launch of its own between the two radius-graph passes
// costs less than the three library launches of cumsum + subtraction it replaces).  A tile is kScanRows rows of 2 x kScanBlock
// entries; thread t owns entries 2t, 2t + 1 of every row, so each of its loads and stores is one coalesced 16-byte lane access
// (a thread owning 16 CONSECUTIVE entries made every wavefront-wide access touch 64 cache lines: 18 us instead of 6).  The sums
// inside a tile are 32-bit (an entry is the edge count of ONE atom: < 2^13 at the largest structure the radius graph takes,
// 16 384 of them < 2^27), the carry between tiles 64-bit.
constexpr int kScanBlock = 1024, kScanRows = 8, kScanWaves = kScanBlock / kWave;

// inclusive prefix sum over the 64 lanes with DPP adds (no LDS): within rows of 16 lanes, then lane 15 of rows 0 / 2 into rows
// 1 / 3, then lane 31 into the upper half
__device__ __forceinline__ int wave_inclusive_scan(int x)
{
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);      // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);      // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);      // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);      // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return x;
}

__global__ __launch_bounds__(kScanBlock) void offsets_scan_kernel(const int64_t* __restrict__ counts, int64_t n,
                                                                  int64_t* __restrict__ offsets, int64_t* __restrict__ total)
{
    // per (row, wavefront) sums of a tile, then their exclusive prefix in row-major order; [parity of the tile] so that a
    // wavefront one barrier ahead does not write what a slower one still reads
    __shared__ int sums[2][kScanRows * kScanWaves + 1];
    const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
    int64_t carry = 0;
    int parity = 0;
    for (int64_t base = 0; base < n; base += (int64_t)kScanRows * 2 * kScanBlock, parity ^= 1) {
        int* tile_sums = sums[parity];
        int v0[kScanRows], v1[kScanRows], incl[kScanRows];
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {
            const int64_t i = base + ((int64_t)r * kScanBlock + threadIdx.x) * 2;
            if (i + 1 < n) {
                const longlong2 pair = *reinterpret_cast<const longlong2*>(counts + i);
                v0[r] = (int)pair.x;
                v1[r] = (int)pair.y;
            } else {
                v0[r] = i < n ? (int)counts[i] : 0;
                v1[r] = 0;
            }
        }
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {
            incl[r] = wave_inclusive_scan(v0[r] + v1[r]);
            if (lane == kWave - 1) tile_sums[r * kScanWaves + wave] = incl[r];
        }
        __syncthreads();
        if (wave == 0) {
            // 128 sums, two per lane, in row-major order -> what lies before each of them in the tile; the tile's total behind
            static_assert(kScanRows * kScanWaves == 2 * kWave, "one wavefront scans the tile's sums two per lane");
            const int a = tile_sums[2 * lane], b = tile_sums[2 * lane + 1];
            const int through = wave_inclusive_scan(a + b);
            tile_sums[2 * lane] = through - a - b;
            tile_sums[2 * lane + 1] = through - b;
            if (lane == kWave - 1) tile_sums[kScanRows * kScanWaves] = through;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {
            const int64_t i = base + ((int64_t)r * kScanBlock + threadIdx.x) * 2;
            const int64_t first = carry + (tile_sums[r * kScanWaves + wave] + (incl[r] - v0[r] - v1[r]));
            if (i + 1 < n) {
                longlong2 pair;
                pair.x = first;
                pair.y = first + v0[r];
                *reinterpret_cast<longlong2*>(offsets + i) = pair;
            } else if (i < n) {
                offsets[i] = first;
            }
        }
        carry += tile_sums[kScanRows * kScanWaves];
    }
    if (threadIdx.x == 0) *total = carry;
}

// ---------------------------------------------------------------------------------------------------------------
// N1 as the EGNN score network builds it, in TWO launches: hit masks, then emission
// ---------------------------------------------------------------------------------------------------------------
// count -> scan -> fill evaluates every pair twice and puts a one-workgroup scan of B*N counts between two chip-wide launches.  Here
// the adjacency of a structure is kept as what the ballot already is -- one 64-bit word per (source row, 64 destinations) -- in a
// caller's workspace (N = 64: 512 bytes per structure); the second launch needs no positions and no arithmetic: it sums the totals
// of the structures before its own (B words), scans its N row counts, and turns the words into ordered 16-byte pairs.
// One workgroup per structure in both; a wavefront owns a CONTIGUOUS run of source rows, so its edges are one contiguous run of the
// list and the write position is a running sum.  Same pair test as radius_graph_kernel (images_within_cutoff): same edges, same order.
constexpr int kGraphMaxAtoms = 1024, kGraphMaxBatch = 2048;

template <int THREADS>
__global__ __launch_bounds__(THREADS) void egnn_graph_mask_kernel(const float* __restrict__ relative, const float* __restrict__ lattice,
                                                                   int lattice_stride, float clip_min, float rc, int N,
                                                                   int64_t* __restrict__ counts, unsigned long long* __restrict__ masks,
                                                                   int64_t* __restrict__ totals, uint32_t* status)
{
    extern __shared__ float lds[];
    float* pos = lds;                                    // [N][3]
    float* lv = lds + 3 * N;                             // [27][3]
    int* wave_total = reinterpret_cast<int*>(lv + 81);   // [THREADS / kWave]
    const int64_t b = blockIdx.x;
    const float* P = relative + b * N * 3;
    float cl[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) cl[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = lattice[b * lattice_stride + k];
        cl[4 * k] = v < clip_min ? clip_min : v;             // torch.clip(min=): a NaN stays a NaN
    }
    for (int i = threadIdx.x; i < 3 * N; i += THREADS) {
        const int c = i % 3;
        pos[i] = P[i] * (c == 0 ? cl[0] : c == 1 ? cl[4] : cl[8]);
    }
    if (threadIdx.x < 81) {
        const int l = threadIdx.x / 3, c = threadIdx.x % 3;
        const float rel[3] = {(float)(l / 9 - 1), (float)((l / 3) % 3 - 1), (float)(l % 3 - 1)};
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) acc = __builtin_fmaf(rel[k], c == 0 ? cl[k * 3] : c == 1 ? cl[k * 3 + 1] : cl[k * 3 + 2], acc);
        lv[threadIdx.x] = acc;
    }
    if (threadIdx.x == 96 && status) {
        if (!(crossing_distance(cl) > rc)) atomicOr(status, MDX_STATUS_CUTOFF_TOO_LARGE);
    }
    const bool ortho = cl[0] > 0.0f && cl[4] > 0.0f && cl[8] > 0.0f && rc * 2.2f <= fminf(cl[0], fminf(cl[4], cl[8]));
    const float inv_lx = ortho ? 1.0f / cl[0] : 0.0f, inv_ly = ortho ? 1.0f / cl[4] : 0.0f, inv_lz = ortho ? 1.0f / cl[8] : 0.0f;
    __syncthreads();
    const float rc2 = rc * rc;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int words = (N + kWave - 1) / kWave;                           // per source row
    const int rows = (N + THREADS / kWave - 1) / (THREADS / kWave);      // per wavefront, contiguous
    const int first = wave * rows, last = min(N, first + rows);
    int total = 0;
    if (N <= kWave) {
        // one word per row: the lane's destination atom stays in registers over the wavefront's rows
        const int j = lane < N ? lane : 0;
        const float pjx = pos[3 * j], pjy = pos[3 * j + 1], pjz = pos[3 * j + 2];
        for (int i = first; i < last; ++i) {
            const bool hit = any_image_within_cutoff(ortho, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], pjx, pjy, pjz, inv_lx, inv_ly,
                                                     inv_lz, cl[0], cl[4], cl[8], lv, rc2);
            const unsigned long long hits = __ballot(lane < N && hit);
            const int count = __popcll(hits);
            if (lane == 0) {
                masks[b * N + i] = hits;
                counts[b * N + i] = count;
            }
            total += count;
        }
    } else
    for (int i = first; i < last; ++i) {
        const float pix = pos[3 * i], piy = pos[3 * i + 1], piz = pos[3 * i + 2];
        unsigned long long* row_words = masks + (b * N + i) * words;
        int count = 0;
        for (int w = 0; w < words; ++w) {
            const int j = w * kWave + lane;
            bool hit = false;
            if (j < N)
                hit = any_image_within_cutoff(ortho, pix, piy, piz, pos[3 * j], pos[3 * j + 1], pos[3 * j + 2], inv_lx, inv_ly,
                                              inv_lz, cl[0], cl[4], cl[8], lv, rc2);
            const unsigned long long hits = __ballot(hit);
            if (lane == 0) row_words[w] = hits;
            count += __popcll(hits);
        }
        if (lane == 0) counts[b * N + i] = count;
        total += count;
    }
    if (lane == 0) wave_total[wave] = total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int64_t sum = 0;
        for (int w = 0; w < THREADS / kWave; ++w) sum += wave_total[w];
        totals[b] = sum;
    }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void egnn_graph_emit_kernel(int N, int64_t B, const int64_t* __restrict__ counts,
                                                                   const unsigned long long* __restrict__ masks,
                                                                   const int64_t* __restrict__ totals, int64_t* __restrict__ offsets,
                                                                   int64_t* __restrict__ n_edges, int64_t* __restrict__ edges,
                                                                   int64_t capacity, uint32_t* status)
{
    extern __shared__ int row_offset[];                  // [N + 1]: exclusive scan of the structure's row counts, the total behind
    __shared__ int64_t partial[THREADS / kWave];
    const int64_t b = blockIdx.x;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
    const int words = (N + kWave - 1) / kWave;
    const int rows = (N + THREADS / kWave - 1) / (THREADS / kWave);
    const int first = min(N, wave * rows), last = min(N, first + rows);
    const int n_words = (last - first) * words;          // this wavefront's hit words: one contiguous run
    const unsigned long long* my_words = masks + (b * N + first) * words;
    unsigned long long held = lane < n_words ? my_words[lane] : 0ull;        // requested before anything waits
    // edges of the structures before this one
    int64_t before = 0;
    for (int64_t k = threadIdx.x; k < b; k += THREADS) before += totals[k];
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) before += __shfl_xor(before, o, kWave);
    if (lane == 0) partial[wave] = before;
    if (wave == 0) {
        int carry = 0;
        for (int c0 = 0; c0 < N; c0 += kWave) {
            const int i = c0 + lane;
            const int v = i < N ? (int)counts[b * N + i] : 0;
            const int incl = wave_inclusive_scan(v);
            if (i < N) row_offset[i] = carry + incl - v;
            carry += __shfl(incl, kWave - 1, kWave);
        }
        if (lane == 0) row_offset[N] = carry;
    }
    __syncthreads();
    int64_t base = 0;
#pragma unroll
    for (int w = 0; w < THREADS / kWave; ++w) base += partial[w];
    for (int i = threadIdx.x; i < N; i += THREADS) offsets[b * N + i] = base + row_offset[i];
    const int total = row_offset[N];
    if (threadIdx.x == 0) {
        if (b == B - 1) *n_edges = base + total;
        // a caller-sized edge list that is too small: nothing is written beyond it, and the caller is told
        if (status && base + total > capacity) atomicOr(status, MDX_STATUS_GRAPH_CAPACITY);
    }
    if (n_words == 0) return;
    int64_t e = base + row_offset[first];
    int64_t src = b * N + first;
    int w_in_row = 0;
    for (int c0 = 0; c0 < n_words; c0 += kWave) {
        if (c0) held = c0 + lane < n_words ? my_words[c0 + lane] : 0ull;
        const int limit = min(kWave, n_words - c0);
        for (int k = 0; k < limit; ++k) {
            const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)held, k), hi = __builtin_amdgcn_readlane((uint32_t)(held >> 32), k);
            const unsigned long long hits = ((unsigned long long)hi << 32) | lo;
            if ((hits >> lane) & 1ull) {
                const int64_t at = e + __popcll(hits & ((1ull << lane) - 1ull));
                longlong2 pair;
                pair.x = src;
                pair.y = b * N + w_in_row * kWave + lane;
                if (at < capacity) reinterpret_cast<longlong2*>(edges)[at] = pair;        // one 16-B store per edge
            }
            e += __popcll(hits);
            if (++w_in_row == words) { w_in_row = 0; ++src; }
        }
    }
}

template <int THREADS>
static void launch_graph_two_pass(const float* relative_coordinates, const float* lattice_parameters, int lattice_stride, float clip_min,
                                  float rc, int64_t batch, int N, int64_t capacity, int64_t* counts, int64_t* offsets,
                                  int64_t* n_edges, int64_t* edges_out, uint32_t* status, uint64_t* workspace, hipStream_t stream)
{
    unsigned long long* masks = reinterpret_cast<unsigned long long*>(workspace);
    int64_t* totals = reinterpret_cast<int64_t*>(workspace) + batch * N * (int64_t)cdiv(N, kWave);
    const size_t lds = sizeof(float) * (3 * (size_t)N + 81) + sizeof(int) * (THREADS / kWave);
    hipLaunchKernelGGL(egnn_graph_mask_kernel<THREADS>, dim3((unsigned)batch), dim3(THREADS), lds, stream, relative_coordinates,
                       lattice_parameters, lattice_stride, clip_min, rc, N, counts, masks, totals, status);
    hipLaunchKernelGGL(egnn_graph_emit_kernel<THREADS>, dim3((unsigned)batch), dim3(THREADS), sizeof(int) * ((size_t)N + 1), stream,
                       N, batch, (const int64_t*)counts, (const unsigned long long*)masks, (const int64_t*)totals, offsets, n_edges,
                       edges_out, capacity, status);
}

// ---------------------------------------------------------------------------------------------------------------
// RNG fills and math probes
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void rng_fill_kernel(int kind, uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag,
                                                          int64_t n_items, int width, float* out)
{
    const int subs = (width + 3) / 4;
    const int64_t total = n_items * subs;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t it = t / subs;
        const int sub = (int)(t - it * subs);
        const u32x4 r = philox4x32_10((uint32_t)it, (call << 8) | (uint32_t)sub, draw, tag, k0, k1);
        float v[4];
        if (kind == 1) {
            box_muller(r.v[0], r.v[1], v[0], v[1]);
            box_muller(r.v[2], r.v[3], v[2], v[3]);
        } else {
            for (int l = 0; l < 4; ++l) {
                const float u = u01(r.v[l]);
                v[l] = (kind == 2) ? gumbel_from_u(u) : u;
            }
        }
        for (int l = 0; l < 4 && sub * 4 + l < width; ++l) out[it * width + sub * 4 + l] = v[l];
    }
}

__global__ __launch_bounds__(kBlock) void math_probe_kernel(int fn, const float* x, int64_t count, float* y)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        float r;
        if (fn == 0) r = logf_(v);
        else if (fn == 1) r = expf_(v);
        else {
            float s, c;
            sincospif_(v, s, c);
            r = (fn == 2) ? s : c;
        }
        y[i] = r;
    }
}

inline unsigned flat_grid(int64_t work_items)
{
    int64_t blocks = cdiv(work_items, kBlock);
    if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 resident blocks; the rest is grid-strided
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

template <bool LDS_WEIGHTS>
static void launch_mlp_sampler(int G, unsigned grid, size_t lds, hipStream_t st, const MlpSampleArgs& a)
{
    const dim3 block(kMlpWaves * kWave);
    switch (G) {
        case 1: hipLaunchKernelGGL((mlp_pc_sample_kernel<1, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        case 2: hipLaunchKernelGGL((mlp_pc_sample_kernel<2, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        case 4: hipLaunchKernelGGL((mlp_pc_sample_kernel<4, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        case 8: hipLaunchKernelGGL((mlp_pc_sample_kernel<8, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        case 16: hipLaunchKernelGGL((mlp_pc_sample_kernel<16, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        case 32: hipLaunchKernelGGL((mlp_pc_sample_kernel<32, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
        default: hipLaunchKernelGGL((mlp_pc_sample_kernel<64, LDS_WEIGHTS, 0>), dim3(grid), block, lds, st, a); break;
    }
}

#define MDX_FOLDED_FAMILY(X) X(122) X(123) X(124) X(132) X(133) X(134)
#define MDX_PADDED_FAMILY(X) X(212) X(213) X(214) X(222) X(223) X(224) X(232) X(233) X(234)

static const void* mlp_sampler_lds_function(int G, int spec)
{
#define MDX_CASE(S) if (spec == S) return (const void*)mlp_pc_sample_kernel<8, true, S>;
    MDX_FOLDED_FAMILY(MDX_CASE)
    MDX_PADDED_FAMILY(MDX_CASE)
#undef MDX_CASE
    if (spec == 1) return (const void*)mlp_pc_sample_kernel<8, true, 1>;
    switch (G) {
        case 1: return (const void*)mlp_pc_sample_kernel<1, true, 0>;
        case 2: return (const void*)mlp_pc_sample_kernel<2, true, 0>;
        case 4: return (const void*)mlp_pc_sample_kernel<4, true, 0>;
        case 8: return (const void*)mlp_pc_sample_kernel<8, true, 0>;
        case 16: return (const void*)mlp_pc_sample_kernel<16, true, 0>;
        case 32: return (const void*)mlp_pc_sample_kernel<32, true, 0>;
        default: return (const void*)mlp_pc_sample_kernel<64, true, 0>;
    }
}

// =================================================================================================================
// C ABI
// =================================================================================================================
extern "C" {

int mdx_abi_version(void) { return MDX_ABI_VERSION; }

const char* mdx_status_string(int status)
{
    switch (status) {
        case MDX_OK: return "ok";
        case MDX_ERR_INVALID_ARG: return "invalid argument";
        case MDX_ERR_UNSUPPORTED: return "unsupported size or option";
        case MDX_ERR_HIP: return "HIP runtime error at launch";
        default: return "unknown status";
    }
}

int mdx_noise_schedule_build(int T, int schedule_type, double time_delta, double sigma_min, double sigma_max,
                             double corrector_step_epsilon, int C, float* time, float* sigma, float* sigma_squared,
                             float* g, float* g_squared, float* epsilon, float* sqrt_2_epsilon, float* beta,
                             float* alpha_bar, float* q_matrix, float* q_bar_matrix, float* q_bar_tm1_matrix,
                             mdx_stream_t stream)
{
    if (T < 2 || C < 2 || (schedule_type != 0 && schedule_type != 1)) return MDX_ERR_INVALID_ARG;
    if (C > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;      // qbar_chain keeps one row in MDX_MAX_CLASSES registers
    if (!time || !sigma || !sigma_squared || !g || !g_squared || !epsilon || !sqrt_2_epsilon || !beta || !alpha_bar ||
        !q_matrix || !q_bar_matrix || !q_bar_tm1_matrix)
        return MDX_ERR_INVALID_ARG;
    ScheduleArgs a{T, schedule_type, C, time_delta, sigma_min, sigma_max, corrector_step_epsilon, time, sigma,
                   sigma_squared, g, g_squared, epsilon, sqrt_2_epsilon, beta, alpha_bar, q_matrix, q_bar_matrix,
                   q_bar_tm1_matrix};
    hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), a);
    return launch_status();
}

int mdx_index_set(int32_t* d_index, int32_t value, mdx_stream_t stream)
{
    if (!d_index) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(index_kernel, dim3(1), dim3(1), 0, as_stream(stream), d_index, value, 0);
    return launch_status();
}

int mdx_index_add(int32_t* d_index, int32_t delta, mdx_stream_t stream)
{
    if (!d_index) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(index_kernel, dim3(1), dim3(1), 0, as_stream(stream), d_index, delta, 1);
    return launch_status();
}

static int check_index(const mdx_schedule_t* s, int mode, int index_i, const int32_t* d_index)
{
    if (!s || (mode != MDX_PREDICTOR && mode != MDX_CORRECTOR)) return MDX_ERR_INVALID_ARG;
    if (!d_index) {   // with a device-resident index the caller guarantees the range
        if (mode == MDX_PREDICTOR && (index_i < 1 || index_i > s->total_time_steps)) return MDX_ERR_INVALID_ARG;
        if (mode == MDX_CORRECTOR && (index_i < 0 || index_i > s->total_time_steps - 1)) return MDX_ERR_INVALID_ARG;
    }
    return MDX_OK;
}

int mdx_fill_time_sigma(const mdx_schedule_t* sched_host, int mode, int index_i, const int32_t* d_index,
                        float* time_out, float* sigma_out, int64_t batch, mdx_stream_t stream)
{
    const int rc = check_index(sched_host, mode, index_i, d_index);
    if (rc != MDX_OK) return rc;
    if (!time_out || !sigma_out || batch < 0) return MDX_ERR_INVALID_ARG;
    if (batch == 0) return MDX_OK;
    hipLaunchKernelGGL(fill_time_sigma_kernel, dim3(flat_grid(batch)), dim3(kBlock), 0, as_stream(stream),
                       to_dev(sched_host), mode, index_i, d_index, time_out, sigma_out, batch);
    return launch_status();
}

static int coordinates_update(const float* x, const float* s, const float* z, float score_weight, float gaussian_noise_weight,
                              float sigma, const float* weights_dev, int64_t count, float* out, mdx_stream_t stream)
{
    if (count < 0 || (count > 0 && (!x || !s || !z || !out))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    if (aligned16(x) && aligned16(s) && aligned16(z) && aligned16(out))
        hipLaunchKernelGGL(coords_update_kernel<4>, dim3(flat_grid(cdiv(count, 4))), dim3(kBlock), 0, as_stream(stream), x,
                           s, z, score_weight, gaussian_noise_weight, sigma, count, out, weights_dev);
    else
        hipLaunchKernelGGL(coords_update_kernel<1>, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), x, s, z,
                           score_weight, gaussian_noise_weight, sigma, count, out, weights_dev);
    return launch_status();
}

int mdx_relative_coordinates_update(const float* x, const float* s, const float* z, float score_weight,
                                    float gaussian_noise_weight, float sigma, int64_t count, float* out,
                                    mdx_stream_t stream)
{
    return coordinates_update(x, s, z, score_weight, gaussian_noise_weight, sigma, nullptr, count, out, stream);
}

int mdx_relative_coordinates_update_dev(const float* x, const float* s, const float* z, const float* weights, int64_t count,
                                        float* out, mdx_stream_t stream)
{
    if (!weights) return MDX_ERR_INVALID_ARG;
    return coordinates_update(x, s, z, 0.0f, 0.0f, 1.0f, weights, count, out, stream);
}

static int lattice_update(const float* l, const float* s, const float* z, float score_weight, float gaussian_noise_weight,
                          float sigma_n, const float* weights_dev, int64_t count, float* out, mdx_stream_t stream)
{
    if (count < 0 || (count > 0 && (!l || !s || !z || !out))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    hipLaunchKernelGGL(lattice_update_kernel, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), l, s, z,
                       score_weight, gaussian_noise_weight, sigma_n, count, out, weights_dev);
    return launch_status();
}

int mdx_lattice_parameters_update(const float* l, const float* s, const float* z, float score_weight,
                                  float gaussian_noise_weight, float sigma_n, int64_t count, float* out,
                                  mdx_stream_t stream)
{
    return lattice_update(l, s, z, score_weight, gaussian_noise_weight, sigma_n, nullptr, count, out, stream);
}

int mdx_lattice_parameters_update_dev(const float* l, const float* s, const float* z, const float* weights, int64_t count,
                                      float* out, mdx_stream_t stream)
{
    if (!weights) return MDX_ERR_INVALID_ARG;
    return lattice_update(l, s, z, 0.0f, 0.0f, 1.0f, weights, count, out, stream);
}

int mdx_atom_types_update(const float* logits, const int64_t* atom_types, const float* q, const float* q_bar,
                          const float* q_bar_tm1, const float* gumbel, const float* u, int64_t batch,
                          int number_of_atoms, int num_classes, float small_epsilon, int greedy, int one_transition,
                          int64_t* atom_types_out, float* probabilities_out, mdx_stream_t stream)
{
    if (batch < 0 || number_of_atoms < 1) return MDX_ERR_INVALID_ARG;
    if (num_classes < 2) return MDX_ERR_INVALID_ARG;
    if (num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (batch == 0) return MDX_OK;
    if (!logits || !atom_types || !q || !q_bar || !q_bar_tm1 || !gumbel || !atom_types_out || (greedy && !u))
        return MDX_ERR_INVALID_ARG;
    PcArgs a{};
    a.use_tables = 0;
    a.q_explicit = q; a.qbar_explicit = q_bar; a.qbar_tm1_explicit = q_bar_tm1;
    a.greedy = greedy; a.one_transition = one_transition; a.update_types = 1;
    a.small_eps = small_epsilon;
    a.a = atom_types; a.logits = logits; a.gumbel = gumbel; a.u = u;
    a.B = batch; a.N = number_of_atoms; a.C = num_classes; a.d = 3;
    a.a_out = atom_types_out; a.p_out = probabilities_out;
    return launch_pc(a, as_stream(stream));
}

int mdx_pc_step_update(const mdx_schedule_t* sched_host, int mode, int index_i, const int32_t* d_index,
                       const mdx_pc_flags_t* f, const int64_t* atom_types, const float* x, const float* l,
                       const float* logits, const float* score_x, const float* score_l, const float* z_coordinates,
                       const float* gumbel, const float* u, const float* z_lattice, mdx_rng_t rng, int64_t batch,
                       int number_of_atoms, int spatial_dimension, int64_t* atom_types_out, float* x_out, float* l_out,
                       uint32_t* status, mdx_stream_t stream)
{
    const int rc = check_index(sched_host, mode, index_i, d_index);
    if (rc != MDX_OK) return rc;
    if (!f || batch < 0 || number_of_atoms < 1 || spatial_dimension < 1 || spatial_dimension > 3) return MDX_ERR_INVALID_ARG;
    const int C = sched_host->num_classes;
    if (C < 2) return MDX_ERR_INVALID_ARG;
    if (C > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (batch == 0) return MDX_OK;
    if (!x || !score_x || !x_out || !l || !l_out) return MDX_ERR_INVALID_ARG;
    if (f->update_atom_types && (!atom_types || !logits || !atom_types_out)) return MDX_ERR_INVALID_ARG;
    if (!f->use_fixed_lattice_parameters && !score_l) return MDX_ERR_INVALID_ARG;
    if ((int64_t)batch * number_of_atoms > 0xffffffffLL) return MDX_ERR_UNSUPPORTED;   // 32-bit Philox item index
    PcArgs a{};
    a.sched = to_dev(sched_host);
    a.use_tables = 1;
    a.mode = mode; a.index_i = index_i; a.d_index = d_index;
    a.atoms_pow = pow((double)number_of_atoms, 1.0 / (double)spatial_dimension);
    a.greedy = f->atom_type_greedy_sampling; a.one_transition = f->one_atom_type_transition_per_step;
    a.fixed_lattice = f->use_fixed_lattice_parameters; a.update_types = f->update_atom_types;
    a.do_coords = 1; a.do_lattice = 1;
    a.small_eps = f->small_epsilon;
    a.a = atom_types; a.x = x; a.l = l; a.logits = logits; a.score_x = score_x; a.score_l = score_l;
    a.z_coord = z_coordinates; a.gumbel = gumbel; a.u = u; a.z_lat = z_lattice;
    a.rng = rng;
    a.B = batch; a.N = number_of_atoms; a.d = spatial_dimension; a.C = C;
    a.nl = spatial_dimension * (spatial_dimension + 1) / 2;
    a.a_out = atom_types_out; a.x_out = x_out; a.l_out = l_out; a.p_out = nullptr;
    a.status = status;
    return launch_pc(a, as_stream(stream));
}

int mdx_noise_relative_coordinates(const float* x0, const float* z, float sigma, int64_t count, float* out,
                                   mdx_stream_t stream)
{
    if (count < 0 || (count > 0 && (!x0 || !z || !out))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    hipLaunchKernelGGL(noise_coords_kernel, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), x0, z, sigma,
                       count, out);
    return launch_status();
}

int mdx_noise_atom_types(const int64_t* a0, const float* q_bar, const float* u, int64_t n_atoms, int num_classes,
                         int64_t* out, mdx_stream_t stream)
{
    if (n_atoms < 0 || num_classes < 2) return MDX_ERR_INVALID_ARG;
    if (num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (n_atoms == 0) return MDX_OK;
    if (!a0 || !q_bar || !u || !out) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(noise_atom_types_kernel, dim3(flat_grid(n_atoms)), dim3(kBlock), 0, as_stream(stream), a0, q_bar,
                       (int64_t)0, u, n_atoms, num_classes, out);
    return launch_status();
}

int mdx_noise_atom_types_per_atom(const int64_t* a0, const float* q_bar, const float* u, int64_t n_atoms, int num_classes,
                                  int64_t* out, mdx_stream_t stream)
{
    if (n_atoms < 0 || num_classes < 2) return MDX_ERR_INVALID_ARG;
    if (num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (n_atoms == 0) return MDX_OK;
    if (!a0 || !q_bar || !u || !out) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(noise_atom_types_kernel, dim3(flat_grid(n_atoms)), dim3(kBlock), 0, as_stream(stream), a0, q_bar,
                       (int64_t)num_classes * num_classes, u, n_atoms, num_classes, out);
    return launch_status();
}

int mdx_noise_relative_coordinates_sigmas(const float* x0, const float* z, const float* sigmas, int64_t count, float* out,
                                          mdx_stream_t stream)
{
    if (count < 0 || (count > 0 && (!x0 || !z || !sigmas || !out))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    hipLaunchKernelGGL(noise_coords_sigmas_kernel, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), x0, z, sigmas,
                       count, out);
    return launch_status();
}

int mdx_noise_lattice_parameters(const float* l0, const float* z, const float* sigmas_n, int64_t count, float* out,
                                 mdx_stream_t stream)
{
    if (count < 0 || (count > 0 && (!l0 || !z || !sigmas_n || !out))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    hipLaunchKernelGGL(noise_lattice_kernel, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), l0, z, sigmas_n, count,
                       out);
    return launch_status();
}

int mdx_repaint_constrained_rows(const mdx_schedule_t* sched_host, int index_i, const int32_t* d_index,
                                 const float* constrained_x, const int64_t* constrained_a,
                                 const int64_t* constrained_indices, int number_of_constraints, const float* z,
                                 const float* u, mdx_rng_t rng, int64_t batch, int number_of_atoms,
                                 int spatial_dimension, float* x_inout, int64_t* a_inout, mdx_stream_t stream)
{
    if (!sched_host || batch < 0 || number_of_constraints < 0 || number_of_atoms < 1) return MDX_ERR_INVALID_ARG;
    if (spatial_dimension < 1 || spatial_dimension > 3 || number_of_constraints > number_of_atoms) return MDX_ERR_INVALID_ARG;
    if (!d_index && (index_i < 0 || index_i > sched_host->total_time_steps)) return MDX_ERR_INVALID_ARG;
    if (sched_host->num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (batch == 0 || number_of_constraints == 0) return MDX_OK;
    if (!constrained_x || !constrained_a || !constrained_indices || !x_inout || !a_inout) return MDX_ERR_INVALID_ARG;
    RepaintArgs a{};
    a.sched = to_dev(sched_host);
    a.index_i = index_i; a.d_index = d_index;
    a.cx = constrained_x; a.ca = constrained_a; a.cidx = constrained_indices; a.K = number_of_constraints;
    a.z = z; a.u = u; a.rng = rng;
    a.B = batch; a.N = number_of_atoms; a.d = spatial_dimension; a.C = sched_host->num_classes;
    a.x = x_inout; a.a = a_inout;
    hipLaunchKernelGGL(repaint_rows_kernel, dim3(flat_grid(batch * number_of_constraints)), dim3(kBlock), 0,
                       as_stream(stream), a);
    return launch_status();
}

int mdx_forward_diffusion_step(const mdx_schedule_t* sched_host, int index_i, const int32_t* d_index, const float* z,
                               const float* u, mdx_rng_t rng, int64_t batch, int number_of_atoms, int spatial_dimension,
                               float* x_inout, int64_t* a_inout, mdx_stream_t stream)
{
    if (!sched_host || batch < 0 || number_of_atoms < 1) return MDX_ERR_INVALID_ARG;
    if (spatial_dimension < 1 || spatial_dimension > 3) return MDX_ERR_INVALID_ARG;
    if (!d_index && (index_i < 1 || index_i >= sched_host->total_time_steps)) return MDX_ERR_INVALID_ARG;
    if (sched_host->num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (batch == 0) return MDX_OK;
    if (!x_inout || !a_inout) return MDX_ERR_INVALID_ARG;
    ForwardStepArgs a{};
    a.sched = to_dev(sched_host);
    a.index_i = index_i; a.d_index = d_index;
    a.z = z; a.u = u; a.rng = rng;
    a.atoms = batch * number_of_atoms; a.d = spatial_dimension; a.C = sched_host->num_classes;
    a.x = x_inout; a.a = a_inout;
    hipLaunchKernelGGL(forward_step_kernel, dim3(flat_grid(a.atoms)), dim3(kBlock), 0, as_stream(stream), a);
    return launch_status();
}

static int radius_graph_args_ok(const float* cart, const float* cell, float rc, int64_t batch, int N)
{
    if (batch < 0 || N < 1 || !(rc > 0.0f)) return MDX_ERR_INVALID_ARG;
    if (N > 5000) return MDX_ERR_UNSUPPORTED;    // structure tile (12 B per atom) kept under the 64 KiB default dynamic-LDS limit
    if (batch > 0 && (!cart || !cell)) return MDX_ERR_INVALID_ARG;
    return MDX_OK;
}

int mdx_radius_graph_count(const float* cart, const float* cell, float rc, int64_t batch, int N, int unique,
                           int64_t* counts, uint32_t* status, mdx_stream_t stream)
{
    const int ok = radius_graph_args_ok(cart, cell, rc, batch, N);
    if (ok != MDX_OK) return ok;
    if (batch == 0) return MDX_OK;
    if (!counts) return MDX_ERR_INVALID_ARG;
    const int chunks = (int)cdiv(N, kRowsPerBlock);
    const size_t lds = sizeof(float) * (3 * (size_t)N + 81);
    hipLaunchKernelGGL(radius_graph_kernel<false>, dim3((unsigned)(batch * chunks)), dim3(kBlock), lds, as_stream(stream),
                       cart, cell, rc, batch, N, unique, chunks, counts, (const int64_t*)nullptr, (int64_t*)nullptr,
                       (int32_t*)nullptr, (float*)nullptr, status, (int64_t)0, (const float*)nullptr, 0, 0.0f);
    return launch_status();
}

int mdx_radius_graph_fill_capped(const float* cart, const float* cell, float rc, int64_t batch, int N, int unique,
                                 const int64_t* offsets, int64_t capacity, int64_t* edges_out, int32_t* image_out,
                                 float* shifts_out, uint32_t* status, mdx_stream_t stream)
{
    const int ok = radius_graph_args_ok(cart, cell, rc, batch, N);
    if (ok != MDX_OK) return ok;
    if (capacity < 0) return MDX_ERR_INVALID_ARG;
    if (batch == 0) return MDX_OK;
    if (!offsets || (capacity > 0 && !edges_out) || (!unique && capacity > 0 && !image_out)) return MDX_ERR_INVALID_ARG;
    const int chunks = (int)cdiv(N, kRowsPerBlock);
    const size_t lds = sizeof(float) * (3 * (size_t)N + 81);
    hipLaunchKernelGGL(radius_graph_kernel<true>, dim3((unsigned)(batch * chunks)), dim3(kBlock), lds, as_stream(stream),
                       cart, cell, rc, batch, N, unique, chunks, (int64_t*)nullptr, offsets, edges_out, image_out,
                       shifts_out, status, capacity, (const float*)nullptr, 0, 0.0f);
    return launch_status();
}

int64_t mdx_egnn_radius_graph_workspace_words(int64_t batch, int N)
{
    if (batch < 1 || N < 1 || N > kGraphMaxAtoms || batch > kGraphMaxBatch) return 0;
    return batch * N * (int64_t)cdiv(N, kWave) + batch;
}

int mdx_egnn_radius_graph(const float* relative_coordinates, const float* lattice_parameters, int lattice_stride, float clip_min,
                          float rc, int64_t batch, int N, int64_t capacity, int64_t* counts, int64_t* offsets,
                          int64_t* n_edges, int64_t* edges_out, uint32_t* status, uint64_t* workspace, int64_t workspace_words,
                          mdx_stream_t stream)
{
    if (batch < 0 || N < 1 || !(rc > 0.0f) || capacity < 0 || lattice_stride < 3 || !(clip_min >= 0.0f)) return MDX_ERR_INVALID_ARG;
    if (N > 5000) return MDX_ERR_UNSUPPORTED;
    if (!n_edges) return MDX_ERR_INVALID_ARG;
    if (batch == 0) return hipMemsetAsync(n_edges, 0, sizeof(int64_t), as_stream(stream)) == hipSuccess ? MDX_OK : MDX_ERR_HIP;
    if (!relative_coordinates || !lattice_parameters || !counts || !offsets || (capacity > 0 && !edges_out)) return MDX_ERR_INVALID_ARG;
    if (workspace_words < 0 || (workspace_words > 0 && !workspace)) return MDX_ERR_INVALID_ARG;
    const int64_t needed = mdx_egnn_radius_graph_workspace_words(batch, N);
    if (workspace && needed > 0) {
        if (workspace_words < needed) return MDX_ERR_INVALID_ARG;
        // sixteen wavefronts per structure while the launch still fits the chip at once (8 192 wavefront slots), four beyond
        if (N <= 16 || (N <= kWave && batch > 768))
            launch_graph_two_pass<256>(relative_coordinates, lattice_parameters, lattice_stride, clip_min, rc, batch, N, capacity,
                                       counts, offsets, n_edges, edges_out, status, workspace, as_stream(stream));
        else
            launch_graph_two_pass<1024>(relative_coordinates, lattice_parameters, lattice_stride, clip_min, rc, batch, N, capacity,
                                        counts, offsets, n_edges, edges_out, status, workspace, as_stream(stream));
        return launch_status();
    }
    const int chunks = (int)cdiv(N, kRowsPerBlock);
    const size_t lds = sizeof(float) * (3 * (size_t)N + 81);
    const dim3 grid((unsigned)(batch * chunks));
    hipLaunchKernelGGL(radius_graph_kernel<false>, grid, dim3(kBlock), lds, as_stream(stream), relative_coordinates,
                       (const float*)nullptr, rc, batch, N, 1, chunks, counts, (const int64_t*)nullptr, (int64_t*)nullptr,
                       (int32_t*)nullptr, (float*)nullptr, status, (int64_t)0, lattice_parameters, lattice_stride, clip_min);
    hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(kScanBlock), 0, as_stream(stream), (const int64_t*)counts, batch * N,
                       offsets, n_edges);
    hipLaunchKernelGGL(radius_graph_kernel<true>, grid, dim3(kBlock), lds, as_stream(stream), relative_coordinates,
                       (const float*)nullptr, rc, batch, N, 1, chunks, (int64_t*)nullptr, (const int64_t*)offsets, edges_out,
                       (int32_t*)nullptr, (float*)nullptr, status, capacity, lattice_parameters, lattice_stride, clip_min);
    return launch_status();
}

int mdx_radius_graph_fill(const float* cart, const float* cell, float rc, int64_t batch, int N, int unique,
                          const int64_t* offsets, int64_t* edges_out, int32_t* image_out, float* shifts_out,
                          mdx_stream_t stream)
{
    return mdx_radius_graph_fill_capped(cart, cell, rc, batch, N, unique, offsets, INT64_MAX, edges_out, image_out, shifts_out,
                                        nullptr, stream);
}

static int mlp_ok(const mdx_mlp_t* m)
{
    if (!m) return MDX_ERR_INVALID_ARG;
    if (m->number_of_atoms < 1 || m->spatial_dimension < 1 || m->spatial_dimension > 3 || m->num_classes < 2 ||
        m->hidden_size < 1 || m->n_hidden < 1 || m->e_coordinates < 1 || m->e_noise < 1 || m->e_time < 1 ||
        m->e_atom_type < 1 || m->e_lattice < 1)
        return MDX_ERR_INVALID_ARG;
    if (m->n_hidden > MDX_MLP_MAX_HIDDEN || m->num_classes > MDX_MAX_CLASSES) return MDX_ERR_UNSUPPORTED;
    if (!m->w_coordinates_t || !m->b_coordinates || !m->w_noise_t || !m->b_noise || !m->w_time_t || !m->b_time ||
        !m->w_atom_type_t || !m->b_atom_type || !m->w_lattice_t || !m->b_lattice || !m->w_out_a_t || !m->b_out_a ||
        !m->w_out_x_t || !m->b_out_x || !m->w_out_l_t || !m->b_out_l)
        return MDX_ERR_INVALID_ARG;
    for (int k = 0; k < m->n_hidden; ++k)
        if (!m->w_hidden_t[k] || !m->b_hidden[k]) return MDX_ERR_INVALID_ARG;
    return MDX_OK;
}

constexpr size_t kMlpLdsBudget = 64 * 1024;      // default dynamic-LDS limit per workgroup
constexpr int kMaxDevices = 64;                  // per-device bookkeeping of hipFuncSetAttribute opt-ins

int64_t mdx_mlp_image_floats(const mdx_mlp_t* mlp_host)
{
    if (mlp_ok(mlp_host) != MDX_OK) return -1;
    return mlp_offsets(*mlp_host).total;
}

int mdx_mlp_pack_image(const mdx_mlp_t* mlp_host, float* image_out, mdx_stream_t stream)
{
    const int ok = mlp_ok(mlp_host);
    if (ok != MDX_OK) return ok;
    if (!image_out) return MDX_ERR_INVALID_ARG;
    mdx_mlp_t m = *mlp_host;
    m.packed_image = nullptr;
    hipLaunchKernelGGL(mlp_pack_image_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), m, image_out);
    return launch_status();
}

int mdx_mlp_forward(const mdx_mlp_t* mlp_host, const int64_t* atom_types, const float* x, const float* l,
                    const float* time, const float* sigma, int64_t batch, float* logits_out, float* score_x_out,
                    float* score_l_out, mdx_stream_t stream)
{
    const int ok = mlp_ok(mlp_host);
    if (ok != MDX_OK) return ok;
    if (batch < 0) return MDX_ERR_INVALID_ARG;
    if (batch == 0) return MDX_OK;
    if (!atom_types || !x || !l || !time || !sigma || !logits_out || !score_x_out || !score_l_out) return MDX_ERR_INVALID_ARG;
    // x and l are read as broadcast vectors by the layer loops; the forward-only kernel reads them from global memory
    const size_t scratch = sizeof(float) * (size_t)mlp_wave_floats(*mlp_host) * kMlpWaves;
    const size_t image = sizeof(float) * (size_t)mlp_offsets(*mlp_host).total;
    if (scratch > kMlpLdsBudget) return MDX_ERR_UNSUPPORTED;
    const bool in_lds = scratch + image <= 2 * kMlpLdsBudget;
    const int64_t blocks = cdiv(batch, kMlpWaves);
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    if (in_lds && scratch + image > kMlpLdsBudget &&
        hipFuncSetAttribute((const void*)mlp_forward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(scratch + image)) != hipSuccess)
        return MDX_ERR_HIP;
    if (in_lds)
        hipLaunchKernelGGL(mlp_forward_kernel<true>, dim3(grid), dim3(kMlpWaves * kWave), scratch + image, as_stream(stream),
                           *mlp_host, atom_types, x, l, time, sigma, batch, logits_out, score_x_out, score_l_out);
    else
        hipLaunchKernelGGL(mlp_forward_kernel<false>, dim3(grid), dim3(kMlpWaves * kWave), scratch, as_stream(stream),
                           *mlp_host, atom_types, x, l, time, sigma, batch, logits_out, score_x_out, score_l_out);
    return launch_status();
}

int64_t mdx_mlp_pc_sample_workspace_floats(const mdx_mlp_t* mlp_host, int number_of_corrector_steps,
                                           int atom_type_transition_in_corrector, int n_iterations, int64_t batch)
{
    if (!mlp_host || number_of_corrector_steps < 0 || n_iterations < 0 || batch < 0) return -1;
    const int64_t N = mlp_host->number_of_atoms, d = mlp_host->spatial_dimension, C = mlp_host->num_classes;
    const int64_t rec0 = N * (d + C + 1) + kP2Table, rec1 = atom_type_transition_in_corrector ? rec0 : N * d;
    return (rec0 + number_of_corrector_steps * rec1) * batch * n_iterations;
}

// the padded register-resident family (SPEC >= 200): 200 + 10 (first-layer quads / 16) + hidden layers, or 0
inline int padded_family_spec(const mdx_mlp_t& m)
{
    if (!m.folded_padded || m.hidden_size > 64 || m.number_of_atoms > 8 || m.n_hidden < 2 || m.n_hidden > 4) return 0;
    if (mlp_outputs(m) > 64 || mlp_folded_inputs(m) > 192) return 0;
    return 200 + 10 * ((mlp_folded_inputs(m) + 63) / 64) + m.n_hidden;
}

static int mlp_sampler_variant(const mdx_mlp_t& m, uint32_t options)
{
    int spec = matches_template_mlp(m) && !(options & MDX_MLP_SAMPLE_GENERIC_KERNEL) ? 1 : 0;
    const bool plain = options & (MDX_MLP_SAMPLE_GENERIC_KERNEL | MDX_MLP_SAMPLE_UNFOLDED);
    if (!plain && folded_family_spec(m) && !(options & MDX_MLP_SAMPLE_PADDED_FAMILY)) spec = folded_family_spec(m);
    else if (!plain && padded_family_spec(m)) spec = padded_family_spec(m);
    return spec;
}

int mdx_mlp_pc_sample_variant(const mdx_mlp_t* mlp_host, uint32_t options)
{
    if (mlp_ok(mlp_host) != MDX_OK) return -1;
    return mlp_sampler_variant(*mlp_host, options);
}

int mdx_mlp_pc_sample(const mdx_schedule_t* sched_host, const mdx_mlp_t* mlp_host, const mdx_pc_flags_t* f,
                      int number_of_corrector_steps, int atom_type_transition_in_corrector, int start_index,
                      int n_iterations, mdx_rng_t rng, int64_t batch, int64_t* atom_types, float* x, float* l,
                      float* noise_workspace, int64_t workspace_floats, uint32_t options, uint32_t* status,
                      mdx_stream_t stream)
{
    const int ok = mlp_ok(mlp_host);
    if (ok != MDX_OK) return ok;
    if (!sched_host || !f || batch < 0 || number_of_corrector_steps < 0 || n_iterations < 0) return MDX_ERR_INVALID_ARG;
    if (start_index < 1 || start_index > sched_host->total_time_steps || n_iterations > start_index) return MDX_ERR_INVALID_ARG;
    if (sched_host->num_classes != mlp_host->num_classes) return MDX_ERR_INVALID_ARG;
    if (mlp_host->number_of_atoms > kWave) return MDX_ERR_UNSUPPORTED;      // one lane per atom in the update
    if ((int64_t)batch * mlp_host->number_of_atoms > 0xffffffffLL) return MDX_ERR_UNSUPPORTED;
    constexpr uint32_t kKnownOptions = MDX_MLP_SAMPLE_GENERIC_KERNEL | MDX_MLP_SAMPLE_UNFOLDED | MDX_MLP_SAMPLE_CALLER_NOISE |
                                       MDX_MLP_SAMPLE_PADDED_FAMILY |
                                       MDX_MLP_SAMPLE_NO_FIXED_SOFTMAX | MDX_MLP_SAMPLE_NO_P2_TABLE |
                                       MDX_MLP_SAMPLE_DIAG_NO_FORWARD | MDX_MLP_SAMPLE_DIAG_NO_UPDATE;
    if (options & ~kKnownOptions) return MDX_ERR_INVALID_ARG;
#ifndef MDX_DIAGNOSTICS
    if (options & (MDX_MLP_SAMPLE_DIAG_NO_FORWARD | MDX_MLP_SAMPLE_DIAG_NO_UPDATE)) return MDX_ERR_UNSUPPORTED;
#endif
    const bool caller_noise = (options & MDX_MLP_SAMPLE_CALLER_NOISE) != 0;
    if (caller_noise && !noise_workspace) return MDX_ERR_INVALID_ARG;
    if (batch == 0 || n_iterations == 0) return MDX_OK;
    if (!atom_types || !x || !l) return MDX_ERR_INVALID_ARG;
    if (noise_workspace) {
        // as many iterations per launch as the workspace holds (one noise pre-pass + one persistent launch each)
        const int64_t per_iteration = mdx_mlp_pc_sample_workspace_floats(mlp_host, number_of_corrector_steps,
                                                                         atom_type_transition_in_corrector, 1, batch);
        const int64_t fit = workspace_floats / per_iteration;
        if (fit < 1 || (caller_noise && fit < n_iterations)) return MDX_ERR_INVALID_ARG;
        if (fit < n_iterations) {
            for (int done = 0; done < n_iterations;) {
                const int n = (int)(n_iterations - done < fit ? n_iterations - done : fit);
                const int rc = mdx_mlp_pc_sample(sched_host, mlp_host, f, number_of_corrector_steps,
                                                 atom_type_transition_in_corrector, start_index - done, n, rng, batch,
                                                 atom_types, x, l, noise_workspace, workspace_floats, options, status, stream);
                if (rc != MDX_OK) return rc;
                done += n;
            }
            return MDX_OK;
        }
    }
    const size_t per_wave = sizeof(float) * (size_t)mlp_wave_floats(*mlp_host);
    const size_t image = sizeof(float) * (size_t)mlp_offsets(*mlp_host).total;
    if (per_wave * kMlpWaves > kMlpLdsBudget) return MDX_ERR_UNSUPPORTED;
    const bool in_lds = per_wave * kMlpWaves + image <= 2 * kMlpLdsBudget;
    MlpSampleArgs a{};
    PcArgs& pc = a.pc;
    pc.sched = to_dev(sched_host);
    pc.use_tables = 1;
    pc.atoms_pow = pow((double)mlp_host->number_of_atoms, 1.0 / (double)mlp_host->spatial_dimension);
    pc.greedy = f->atom_type_greedy_sampling; pc.one_transition = f->one_atom_type_transition_per_step;
    pc.fixed_lattice = f->use_fixed_lattice_parameters;
    pc.do_coords = 1; pc.do_lattice = 1;
    pc.small_eps = f->small_epsilon;
    pc.rng = rng;
    pc.rng.draw_stride = (uint32_t)number_of_corrector_steps + 1;
    pc.B = batch; pc.N = mlp_host->number_of_atoms; pc.d = mlp_host->spatial_dimension; pc.C = mlp_host->num_classes;
    pc.nl = pc.d * (pc.d + 1) / 2;
    pc.status = status;
    a.mlp = *mlp_host;
    a.M = number_of_corrector_steps;
    a.types_in_corrector = atom_type_transition_in_corrector ? 1 : 0;
    a.start_index = start_index;
    a.n_iterations = n_iterations;
    a.a = atom_types; a.x = x; a.l = l;
    a.rec0 = pc.N * (pc.d + pc.C + 1) + kP2Table;
    a.rec1 = a.types_in_corrector ? a.rec0 : pc.N * pc.d;
    // a record longer than the pre-pass kernel's LDS stage (many correctors x many classes): draw in-kernel instead
    if (noise_workspace && !caller_noise && a.rec0 + (int64_t)a.M * a.rec1 > kStageFloats) noise_workspace = nullptr;
    a.noise = noise_workspace;
    // diag_skip bits of the kernel: 1 no forward, 2 no update (diagnostics builds only), 8 no hoisted softmax, 16 no table
    a.diag_skip = ((options & MDX_MLP_SAMPLE_DIAG_NO_FORWARD) ? 1 : 0) | ((options & MDX_MLP_SAMPLE_DIAG_NO_UPDATE) ? 2 : 0) |
                  ((options & MDX_MLP_SAMPLE_NO_FIXED_SOFTMAX) ? 8 : 0) | ((options & MDX_MLP_SAMPLE_NO_P2_TABLE) ? 16 : 0);
    int G = 1;
    while (G < pc.N) G <<= 1;
    const int64_t blocks = cdiv(batch, kMlpWaves);
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    hipStream_t st = as_stream(stream);
    if (noise_workspace && !caller_noise) {
        NoiseFillArgs nf{};
        nf.rng = pc.rng;
        nf.sched = pc.sched; nf.small_eps = pc.small_eps;
        nf.start_index = start_index; nf.n_iterations = n_iterations;
        nf.types_in_corrector = a.types_in_corrector; nf.greedy = pc.greedy;
        nf.B = batch; nf.N = pc.N; nf.d = pc.d; nf.C = pc.C; nf.rec0 = a.rec0; nf.rec1 = a.rec1; nf.M = a.M;
        nf.out = noise_workspace;
        int per_pass = kBlock / pc.N;
        if (per_pass > kStageFloats / (a.rec0 + a.M * a.rec1)) per_pass = kStageFloats / (a.rec0 + a.M * a.rec1);
        hipLaunchKernelGGL(pc_noise_fill_kernel, dim3(flat_grid(cdiv(batch * n_iterations, per_pass) * kBlock)), dim3(kBlock),
                           0, st, nf);
        if (hipGetLastError() != hipSuccess) return MDX_ERR_HIP;
    }
    if (in_lds) {
        size_t lds = per_wave * kMlpWaves + image;
        // 0: generic instantiation; 1: template dimensions as literals, layer by layer; >= 100: the register-resident family
        // with the folded input / output layers (100 + 10 C + NH)
        const int spec = mlp_sampler_variant(*mlp_host, options);
        // the generic instantiation runs the folded forward when the caller supplied the folded matrices, it pays and fits
        const size_t blobs = sizeof(float) * (((size_t)mlp_folded_in_floats(*mlp_host) + mlp_folded_out_floats(*mlp_host) + 3) & ~(size_t)3);
        if (spec == 0 && !(options & MDX_MLP_SAMPLE_UNFOLDED) && mlp_fold_pays(*mlp_host) && lds + blobs <= 2 * kMlpLdsBudget) {
            a.fold = 1;
            lds += blobs;
        }
        if (lds > kMlpLdsBudget) {       // up to 128 KiB of the CU's 160 KiB: opt in above the 64 KiB default
            // the attribute is a property of the code object: set it when the requirement grows, not on every launch
            // (per device: a process that samples on a second GPU must opt in there as well)
            static std::atomic<size_t> granted[kMaxDevices][24];
            int dev = 0;
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return MDX_ERR_HIP;
            const int slot = spec >= 200 ? 14 + (spec - 210) / 10 * 3 + (spec % 10 - 2)     // 14 .. 22
                             : spec >= 100 ? 8 + (spec - 120) / 10 * 3 + (spec % 10 - 2)    // 8 .. 13
                             : spec ? 7 : (G == 1 ? 0 : G == 2 ? 1 : G == 4 ? 2 : G == 8 ? 3 : G == 16 ? 4 : G == 32 ? 5 : 6);
            if (granted[dev][slot].load() < lds) {
                if (hipFuncSetAttribute(mlp_sampler_lds_function(G, spec), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds) != hipSuccess)
                    return MDX_ERR_HIP;
                granted[dev][slot].store(lds);
            }
        }
#define MDX_CASE(S) \
        if (spec == S) hipLaunchKernelGGL((mlp_pc_sample_kernel<8, true, S>), dim3(grid), dim3(kMlpWaves * kWave), lds, st, a); else
        MDX_FOLDED_FAMILY(MDX_CASE)
        MDX_PADDED_FAMILY(MDX_CASE)
#undef MDX_CASE
        if (spec == 1)
            hipLaunchKernelGGL((mlp_pc_sample_kernel<8, true, 1>), dim3(grid), dim3(kMlpWaves * kWave), lds, st, a);
        else
            launch_mlp_sampler<true>(G, grid, lds, st, a);
    } else {
        launch_mlp_sampler<false>(G, grid, per_wave * kMlpWaves, st, a);
    }
    return launch_status();
}

int mdx_rng_fill(int kind, uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag, int64_t n_items, int width,
                 float* out, mdx_stream_t stream)
{
    if (kind < 0 || kind > 2 || n_items < 0 || width < 1 || width > 1024) return MDX_ERR_INVALID_ARG;
    if (n_items > 0xffffffffLL) return MDX_ERR_UNSUPPORTED;
    if (n_items == 0) return MDX_OK;
    if (!out) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(rng_fill_kernel, dim3(flat_grid(n_items * ((width + 3) / 4))), dim3(kBlock), 0, as_stream(stream),
                       kind, seed, call, draw, tag, n_items, width, out);
    return launch_status();
}

int mdx_math_probe(int fn, const float* x, int64_t count, float* y, mdx_stream_t stream)
{
    if (fn < 0 || fn > 3 || count < 0 || (count > 0 && (!x || !y))) return MDX_ERR_INVALID_ARG;
    if (count == 0) return MDX_OK;
    hipLaunchKernelGGL(math_probe_kernel, dim3(flat_grid(count)), dim3(kBlock), 0, as_stream(stream), fn, x, count, y);
    return launch_status();
}

}  // extern "C"
