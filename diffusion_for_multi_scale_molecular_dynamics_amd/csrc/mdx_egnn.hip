// EGNN helpers behind the C ABI: the passes around the MFMA kernels of mdx_egnn_chain.hip that PyTorch cannot fuse.
// (Rounds 1-4 also wrapped a hipBLASLt GEMM with a bias + SiLU epilogue here, mdx_linear_act, for layer shapes the edge chain
// did not cover; since round 5 every option of the reference's E_GCL runs on the hand-written chain and the library links no
// vendor GEMM -- shapes outside the chain's widths stay plain PyTorch modules on the caller's side.)
//
// mdx_egnn_message_input   first layer of the message MLP (egnn.py:136-160) for an edge list: the reference concatenates
//                     [h_src, h_dst, |dx|^2] per edge and applies Linear(2F+1 -> H); with P = h W_src^T | h W_dst^T
//                     computed per NODE, the per-edge work is out = SiLU(P[src,:H] + P[dst,H:] + b + r w_r): one
//                     bandwidth-bound pass (16 B per lane) instead of two gathers, two adds, an addcmul and a SiLU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mdx_hip.h"

namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ float silu_fast(float a) { return a / (1.0f + __expf(-a)); }

// one thread = 4 consecutive features of one edge; H % 4 == 0
__global__ __launch_bounds__(kBlock) void egnn_message_input_kernel(const float* __restrict__ proj, const int64_t* __restrict__ edges,
                                                                    const float* __restrict__ radial, const float* __restrict__ bias,
                                                                    const float* __restrict__ w_radial, int64_t n_edges, int H,
                                                                    int silu, float* __restrict__ out)
{
    const int quads = H >> 2;
    const int64_t total = n_edges * quads;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = t / quads;
        const int q = (int)(t - e * quads);
        const int64_t src = edges[2 * e], dst = edges[2 * e + 1];
        const float r = radial[e];
        const float4 a = reinterpret_cast<const float4*>(proj + src * 2 * H)[q];
        const float4 b = reinterpret_cast<const float4*>(proj + dst * 2 * H + H)[q];
        const float4 bi = reinterpret_cast<const float4*>(bias)[q];
        const float4 wr = reinterpret_cast<const float4*>(w_radial)[q];
        float4 o;
        o.x = (a.x + b.x + bi.x) + r * wr.x;
        o.y = (a.y + b.y + bi.y) + r * wr.y;
        o.z = (a.z + b.z + bi.z) + r * wr.z;
        o.w = (a.w + b.w + bi.w) + r * wr.w;
        if (silu) { o.x = silu_fast(o.x); o.y = silu_fast(o.y); o.z = silu_fast(o.z); o.w = silu_fast(o.w); }
        reinterpret_cast<float4*>(out + e * H)[q] = o;
    }
}

// EGNNScoreNetwork's per-node inputs in one pass (models/score_networks/egnn_score_network.py:253-281, models/egnn.py:292-330):
//   z[i, 2k] = cos(2 pi x_i . K_k),  z[i, 2k+1] = sin(2 pi x_i . K_k)           the torus uplift of the relative coordinates
//   h[i, :]  = b + sigma_{s(i)} W[:, 0] + W[:, 1 + a_i]                         embedding_in applied to [sigma | one_hot(a_i)]
// (the reference builds [sigma | one_hot] and multiplies by W^T; with a one-hot input that is a column pick -- the same
// binary32 operations in the same order: fl(fl(b + fl(sigma w0)) + w_a), every other term an exact zero).
// (cos kr, sin kr) of the binary32 angle, CORRECTLY ROUNDED: evaluated in binary64 and rounded once.  Why it matters for parity:
// the score is z . Gamma . (z + sum of the layers' translations), a small difference of numbers of magnitude 1, so the rounding
// of every layer's x + trans (half an ulp of 1 = 3e-8 against scores of 2e-3) IS the binary32 noise floor of this network -- and
// two evaluations share that rounding only where their z agree in every bit.  The reference's z are torch's CPU cos / sin
// (<= 1 ulp, 95.5 % of them correctly rounded); ocml's binary32 cosf / sinf differ from those in ~30 % of the entries, which
// alone put every GPU evaluation 1.1e-5 (rel-L2) from the reference's scores instead of 6e-6 (measured: tests/golden/
// net_egnn_c3_wide.npz; 98 k (cos, sin) pairs per forward at C3 -- the cost of binary64 here is nothing).
__device__ __forceinline__ void uplift(float kr, float& c, float& s)
{
    const double a = (double)kr;
    c = (float)cos(a);
    s = (float)sin(a);
}

__global__ __launch_bounds__(kBlock) void egnn_node_inputs_kernel(const float* __restrict__ x, const float* __restrict__ k_vectors,
                                                                  int n_k, const float* __restrict__ sigma, int atoms_per_structure,
                                                                  const int64_t* __restrict__ atom_types,
                                                                  const float* __restrict__ w, const float* __restrict__ b,
                                                                  int F, int H, int64_t n_nodes, float* __restrict__ z,
                                                                  float* __restrict__ h, const float* __restrict__ w2,
                                                                  const float* __restrict__ b2, int H2, float* __restrict__ h2)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    // a second linear map of the same [sigma | one_hot] input (nullable): the first graph layer's per-node projections,
    // with w2 = P W, b2 = P b formed once on the host -- P (W x + b) without the [n_nodes, H] x [H, 2H] product
    if (h2) {
        for (int64_t t = t0; t < n_nodes * H2; t += stride) {
            const int64_t i = t / H2;
            const int j = (int)(t - i * H2);
            const float s = sigma[i / atoms_per_structure];
            const int64_t a = atom_types[i];
            float v = b2[j] + s * w2[(int64_t)j * F];
            if (a >= 0 && a + 1 < F) v = v + w2[(int64_t)j * F + 1 + a];
            h2[t] = v;
        }
    }
    for (int64_t t = t0; t < n_nodes * H; t += stride) {
        const int64_t i = t / H;
        const int j = (int)(t - i * H);
        const float s = sigma[i / atoms_per_structure];
        const int64_t a = atom_types[i];
        float v = b[j] + s * w[(int64_t)j * F];
        if (a >= 0 && a + 1 < F) v = v + w[(int64_t)j * F + 1 + a];
        h[t] = v;
    }
    for (int64_t t = t0; t < n_nodes * n_k; t += stride) {
        const int64_t i = t / n_k;
        const int k = (int)(t - i * n_k);
        float kr = 0.0f;
        for (int d = 0; d < 3; ++d) kr = kr + (6.2831855f * x[3 * i + d]) * k_vectors[3 * k + d];
        uplift(kr, z[2 * t], z[2 * t + 1]);
    }
}

// The same with one wavefront per node and 16-byte stores (H, H2 multiples of 4): no index division per element, the node's
// sigma and atom type read once, and both weight matrices staged TRANSPOSED in LDS ([F][H] behind the bias), so that a lane's
// four columns are one 16-byte LDS read instead of eight strided global reads.  Same operations per value, same bits.
__device__ __forceinline__ float4 embed_quad(const float* __restrict__ t, int H, int q, float s, int64_t a, int F)
{
    // t: bias [H], then W^T [F][H]
    const float4 bias = reinterpret_cast<const float4*>(t)[q];
    const float4 w0 = reinterpret_cast<const float4*>(t + H)[q];
    float4 v;
    v.x = bias.x + s * w0.x;
    v.y = bias.y + s * w0.y;
    v.z = bias.z + s * w0.z;
    v.w = bias.w + s * w0.w;
    if (a >= 0 && a + 1 < F) {
        const float4 wa = reinterpret_cast<const float4*>(t + (2 + a) * H)[q];
        v.x = v.x + wa.x;
        v.y = v.y + wa.y;
        v.z = v.z + wa.z;
        v.w = v.w + wa.w;
    }
    return v;
}

__global__ __launch_bounds__(kBlock) void egnn_node_inputs_rows_kernel(const float* __restrict__ x, const float* __restrict__ k_vectors,
                                                                       int n_k, const float* __restrict__ sigma,
                                                                       int atoms_per_structure,
                                                                       const int64_t* __restrict__ atom_types,
                                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                                       int F, int H, int64_t n_nodes, float* __restrict__ z,
                                                                       float* __restrict__ h, const float* __restrict__ w2,
                                                                       const float* __restrict__ b2, int H2,
                                                                       float* __restrict__ h2)
{
    extern __shared__ float tables[];                   // [ (1 + F) H | (1 + F) H2 ]
    float* t1 = tables;
    float* t2 = tables + (1 + F) * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        t1[j] = b[j];
        for (int f = 0; f < F; ++f) t1[(1 + f) * H + j] = w[(int64_t)j * F + f];
    }
    if (h2)
        for (int j = threadIdx.x; j < H2; j += blockDim.x) {
            t2[j] = b2[j];
            for (int f = 0; f < F; ++f) t2[(1 + f) * H2 + j] = w2[(int64_t)j * F + f];
        }
    __syncthreads();
    const int lane = threadIdx.x % 64;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / 64, n_waves = ((int64_t)gridDim.x * blockDim.x) / 64;
    // A pass of a wavefront = `per` consecutive nodes: their torus uplifts first, ONE (node, wave vector) pair per lane -- the
    // uplift is a binary64 sine / cosine (correctly rounded to binary32), a few hundred instructions that 13 lanes of 64 would
    // otherwise run once per node -- then the nodes' embedding rows as 16-byte stores.
    // (at most four nodes per pass: with the 3 wave vectors of one Bloch shell, 64 / n_k = 21 nodes per pass would hand the whole
    // batch to a fifth of the launch's wavefronts)
    const int per = n_k <= 64 ? min(4, 64 / n_k) : 1;
    for (int64_t i0 = wave * per; i0 < n_nodes; i0 += n_waves * per) {
        if (n_k <= 64) {
            const int g = lane / n_k, k = lane - g * n_k;
            const int64_t i = i0 + g;
            if (g < per && i < n_nodes) {
                float kr = 0.0f;
                for (int d = 0; d < 3; ++d) kr = kr + (6.2831855f * x[3 * i + d]) * k_vectors[3 * k + d];
                uplift(kr, z[2 * (i * n_k + k)], z[2 * (i * n_k + k) + 1]);
            }
        } else {
            for (int k = lane; k < n_k; k += 64) {
                float kr = 0.0f;
                for (int d = 0; d < 3; ++d) kr = kr + (6.2831855f * x[3 * i0 + d]) * k_vectors[3 * k + d];
                uplift(kr, z[2 * (i0 * n_k + k)], z[2 * (i0 * n_k + k) + 1]);
            }
        }
        for (int g = 0; g < per && i0 + g < n_nodes; ++g) {
            const int64_t i = i0 + g;
            const float s = sigma[i / atoms_per_structure];
            const int64_t a = atom_types[i];
            if (h2)
                for (int q = lane; q < (H2 >> 2); q += 64) reinterpret_cast<float4*>(h2 + i * H2)[q] = embed_quad(t2, H2, q, s, a, F);
            for (int q = lane; q < (H >> 2); q += 64) reinterpret_cast<float4*>(h + i * H)[q] = embed_quad(t1, H, q, s, a, F);
        }
    }
}

// S^alpha_i = z_i . Gamma^alpha . xhat_i with Gamma^alpha = blockdiag_k(K_k[alpha] [[0,-1],[1,0]])  (egnn_score_network.py:103-133,
// 283-290): per node and direction, sum_k K_k[alpha] (z_{2k+1} xhat_{2k} - z_{2k} xhat_{2k+1}).
__global__ __launch_bounds__(kBlock) void egnn_scores_kernel(const float* __restrict__ z, const float* __restrict__ x_hat,
                                                             const float* __restrict__ k_vectors, int n_k, int64_t n_nodes,
                                                             float* __restrict__ scores)
{
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_nodes * 3; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / 3;
        const int alpha = (int)(t - 3 * i);
        const float* zi = z + i * 2 * n_k;
        const float* xi = x_hat + i * 2 * n_k;
        float acc = 0.0f;
        for (int k = 0; k < n_k; ++k) {
            const float kk = k_vectors[3 * k + alpha];
            acc = acc + ((zi[2 * k] * -kk) * xi[2 * k + 1] + (zi[2 * k + 1] * kk) * xi[2 * k]);
        }
        scores[t] = acc;
    }
}

// Everything EGNNScoreNetwork computes per node behind the last graph layer, in one launch (one wavefront per node):
//   logits[i, c] = h_i . W_c + b_c          EGNN.node_classification_layer (models/egnn.py:362-385), with the MASK class's
//                                           logit set to -inf (score_network.py:183-185) when mask_class >= 0
//   scores[i, :]                            as egnn_scores_kernel
//   zero_out[0 .. n_zero)                   the all-zero lattice output of the network (egnn_score_network.py:299-303)
// A row of h is read once as 16-byte lane loads; the C <= kMaxClasses dot products share it and end in a butterfly.
// sum over the 64 lanes with DPP adds (rows of 16, then lane 15 / 31 of the rows below; the total is lane 63's), handed to every
// lane through a scalar register: no LDS permutes, a fixed order
__device__ __forceinline__ float wave_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

constexpr int kMaxClasses = 8;
constexpr int kWaveSize = 64;

__global__ __launch_bounds__(kBlock) void egnn_outputs_kernel(const float* __restrict__ z, const float* __restrict__ x_hat,
                                                              const float* __restrict__ k_vectors, int n_k,
                                                              const float* __restrict__ h, const float* __restrict__ cw,
                                                              const float* __restrict__ cb, int H, int C, int mask_class,
                                                              int64_t n_nodes, float* __restrict__ scores,
                                                              float* __restrict__ logits, float* __restrict__ zero_out,
                                                              int64_t n_zero)
{
    const int64_t tid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, n_threads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = tid; t < n_zero; t += n_threads) zero_out[t] = 0.0f;
    const int lane = threadIdx.x % kWaveSize;
    const int quads = H >> 2;
    // kNodesPerWave nodes per wavefront and pass: their rows of h, and the score operands of 3 lanes per node, are all
    // requested before anything is reduced (one row per pass left most of the launch waiting on a single 1 KB read)
    constexpr int kNodesPerWave = 4;
    const int64_t n_waves = n_threads / kWaveSize;
    for (int64_t node0 = (tid / kWaveSize) * kNodesPerWave; node0 < n_nodes; node0 += n_waves * kNodesPerWave) {
        float acc = 0.0f;
        const int score_node = (lane - (kWaveSize - 3 * kNodesPerWave)) / 3, alpha = (lane - (kWaveSize - 3 * kNodesPerWave)) % 3;
        const bool scores_lane = lane >= kWaveSize - 3 * kNodesPerWave && node0 + score_node < n_nodes;
        if (scores_lane) {
            const float* zi = z + (node0 + score_node) * 2 * n_k;
            const float* xi = x_hat + (node0 + score_node) * 2 * n_k;
            for (int k = 0; k < n_k; ++k) {
                const float kk = k_vectors[3 * k + alpha];
                acc = acc + ((zi[2 * k] * -kk) * xi[2 * k + 1] + (zi[2 * k + 1] * kk) * xi[2 * k]);
            }
        }
        float part[kNodesPerWave][kMaxClasses];
#pragma unroll
        for (int m = 0; m < kNodesPerWave; ++m)
#pragma unroll
            for (int c = 0; c < kMaxClasses; ++c) part[m][c] = 0.0f;
        for (int q = lane; q < quads; q += kWaveSize) {
            float4 hv[kNodesPerWave];
#pragma unroll
            for (int m = 0; m < kNodesPerWave; ++m)
                hv[m] = node0 + m < n_nodes ? reinterpret_cast<const float4*>(h + (node0 + m) * H)[q] : float4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < kMaxClasses; ++c) {
                if (c < C && c != mask_class) {
                    const float4 wv = reinterpret_cast<const float4*>(cw + (int64_t)c * H)[q];
#pragma unroll
                    for (int m = 0; m < kNodesPerWave; ++m)
                        part[m][c] += (hv[m].x * wv.x + hv[m].y * wv.y) + (hv[m].z * wv.z + hv[m].w * wv.w);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < kMaxClasses; ++c) {
            if (c < C) {
#pragma unroll
                for (int m = 0; m < kNodesPerWave; ++m) {
                    float v = part[m][c];
                    if (c != mask_class) v = wave_sum(v);
                    if (lane == 0 && node0 + m < n_nodes) logits[(node0 + m) * C + c] = (c == mask_class) ? -__builtin_inff() : v + cb[c];
                }
            }
        }
        if (scores_lane) scores[(node0 + score_node) * 3 + alpha] = acc;
    }
}

// Segment kernels on the radius graph's sorted edge list (edges of node i are rows [offset_i, offset_i + degree_i)):
// one wavefront per node, a row of H floats read as 16-byte lane loads (H = 256: one fully coalesced 1-KB row per
// instruction), four rows in flight.  No atomics; the summation order is fixed (edge order), run-to-run deterministic.
constexpr int kWave = 64;

// trans[i, :] = scale_i * sum_e coord_diff[e, :] * (hidden[e, :] . w)      -- last layer of E_GCL.coord_model
// (Linear(H, 1, bias=False), models/egnn.py:162-200) + the multiplication with coord_diff + the segment sum/mean
__global__ __launch_bounds__(kBlock) void egnn_coord_head_kernel(const float* __restrict__ hidden, const float* __restrict__ w,
                                                                 const float* __restrict__ coord_diff,
                                                                 const int64_t* __restrict__ offsets,
                                                                 const int64_t* __restrict__ degree, int64_t n_nodes, int H,
                                                                 int d, int mean, float* __restrict__ trans)
{
    const int lane = threadIdx.x % kWave;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
    const int quads = H >> 2;
    for (int64_t node = wave; node < n_nodes; node += n_waves) {
        const int64_t e0 = offsets[node], deg = degree[node];
        float acc = 0.0f;                                   // lane k < d: component k of the node's translation
        auto row_dot = [&](int64_t e) {                     // this lane's share of hidden[e, :] . w
            float part = 0.0f;
            for (int q = lane; q < quads; q += kWave) {
                const float4 hv = reinterpret_cast<const float4*>(hidden + e * H)[q];
                const float4 wv = reinterpret_cast<const float4*>(w)[q];
                part += (hv.x * wv.x + hv.y * wv.y) + (hv.z * wv.z + hv.w * wv.w);
            }
            return part;
        };
        auto finish = [&](int64_t e, float part) {
#pragma unroll
            for (int o = kWave / 2; o > 0; o >>= 1) part += __shfl_xor(part, o, kWave);
            if (lane < d) acc += coord_diff[e * d + lane] * part;
        };
        int64_t e = e0;
        for (; e + 4 <= e0 + deg; e += 4) {                 // four rows in flight; edges still accumulated in order
            const float p0 = row_dot(e), p1 = row_dot(e + 1), p2 = row_dot(e + 2), p3 = row_dot(e + 3);
            finish(e, p0); finish(e + 1, p1); finish(e + 2, p2); finish(e + 3, p3);
        }
        for (; e < e0 + deg; ++e) finish(e, row_dot(e));
        if (lane < d) trans[node * d + lane] = (mean && deg > 0) ? acc / (float)deg : acc;      // (true division: egnn_utils.py:66-68)
    }
}

// out[i, :] = scale_i * sum_e data[e, :]                                     -- unsorted_segment_sum/mean of the messages
// (models/egnn_utils.py:11-70) for sorted segments
__global__ __launch_bounds__(kBlock) void segment_rows_kernel(const float* __restrict__ data, const int64_t* __restrict__ offsets,
                                                              const int64_t* __restrict__ degree, int64_t n_nodes, int H, int mean,
                                                              float* __restrict__ out)
{
    const int lane = threadIdx.x % kWave;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
    const int quads = H >> 2;
    for (int64_t node = wave; node < n_nodes; node += n_waves) {
        const int64_t e0 = offsets[node], deg = degree[node];
        const float count = (mean && deg > 0) ? (float)deg : 1.0f;
        for (int q = lane; q < quads; q += kWave) {
            float4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
            for (int64_t e = e0; e < e0 + deg; ++e) {
                const float4 v = reinterpret_cast<const float4*>(data + e * H)[q];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            if (mean) { acc.x /= count; acc.y /= count; acc.z /= count; acc.w /= count; }
            reinterpret_cast<float4*>(out + node * H)[q] = acc;
        }
    }
}

}  // namespace

extern "C" {

int mdx_egnn_message_input(const float* node_proj, const int64_t* edges, const float* radial, const float* bias,
                           const float* w_radial, int64_t n_edges, int H, int silu, float* out, mdx_stream_t stream)
{
    if (n_edges < 0 || H < 4 || (H & 3)) return H >= 1 && (H & 3) ? MDX_ERR_UNSUPPORTED : MDX_ERR_INVALID_ARG;
    if (n_edges == 0) return MDX_OK;
    if (!node_proj || !edges || !radial || !bias || !w_radial || !out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_edges * (H >> 2) + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_message_input_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       node_proj, edges, radial, bias, w_radial, n_edges, H, silu, out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

static unsigned node_grid(int64_t n_nodes)
{
    int64_t blocks = (n_nodes * kWave + kBlock - 1) / kBlock;       // one wavefront per node
    if (blocks > 16384) blocks = 16384;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

int mdx_egnn_node_inputs(const float* x, const float* k_vectors, int n_k, const float* sigma, int atoms_per_structure,
                         const int64_t* atom_types, const float* emb_weight, const float* emb_bias, int n_features, int H,
                         int64_t n_nodes, float* z_out, float* h_out, const float* second_weight, const float* second_bias,
                         int second_width, float* second_out, mdx_stream_t stream)
{
    if (n_nodes < 0 || n_k < 1 || atoms_per_structure < 1 || n_features < 2 || H < 1) return MDX_ERR_INVALID_ARG;
    if (n_nodes == 0) return MDX_OK;
    if (!x || !k_vectors || !sigma || !atom_types || !emb_weight || !emb_bias || !z_out || !h_out) return MDX_ERR_INVALID_ARG;
    if (second_out && (!second_weight || !second_bias || second_width < 1)) return MDX_ERR_INVALID_ARG;
    const size_t table_bytes = sizeof(float) * (size_t)(1 + n_features) * ((size_t)H + (second_out ? (size_t)second_width : 0));
    if ((H & 3) == 0 && (!second_out || (second_width & 3) == 0) && table_bytes <= 48 * 1024) {
        // a workgroup stages the tables once (<= 48 KB from L2) and then only stores: eight wavefronts per SIMD keep enough 16-byte
        // stores in flight (four nodes per wavefront at C3; sixteen per wavefront left the chip a quarter occupied: 41 us)
        int64_t blocks = (n_nodes + 15) / 16;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(egnn_node_inputs_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), table_bytes,
                           reinterpret_cast<hipStream_t>(stream), x, k_vectors, n_k, sigma, atoms_per_structure, atom_types,
                           emb_weight, emb_bias, n_features, H, n_nodes, z_out, h_out, second_weight, second_bias, second_width,
                           second_out);
        return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
    }
    int64_t blocks = (n_nodes * H + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_node_inputs_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), x,
                       k_vectors, n_k, sigma, atoms_per_structure, atom_types, emb_weight, emb_bias, n_features, H, n_nodes,
                       z_out, h_out, second_weight, second_bias, second_width, second_out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_scores(const float* z, const float* x_hat, const float* k_vectors, int n_k, int64_t n_nodes, float* scores_out,
                    mdx_stream_t stream)
{
    if (n_nodes < 0 || n_k < 1) return MDX_ERR_INVALID_ARG;
    if (n_nodes == 0) return MDX_OK;
    if (!z || !x_hat || !k_vectors || !scores_out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_nodes * 3 + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_scores_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), z, x_hat,
                       k_vectors, n_k, n_nodes, scores_out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_outputs(const float* z, const float* x_hat, const float* k_vectors, int n_k, const float* h,
                     const float* class_weight, const float* class_bias, int H, int num_classes, int mask_class,
                     int64_t n_nodes, float* scores_out, float* logits_out, float* zero_out, int64_t n_zero,
                     mdx_stream_t stream)
{
    if (n_nodes < 0 || n_k < 1 || H < 4 || num_classes < 1 || n_zero < 0 || mask_class >= num_classes) return MDX_ERR_INVALID_ARG;
    if ((H & 3) || num_classes > kMaxClasses) return MDX_ERR_UNSUPPORTED;
    if (n_zero > 0 && !zero_out) return MDX_ERR_INVALID_ARG;
    if (n_nodes == 0 && n_zero == 0) return MDX_OK;
    if (n_nodes > 0 && (!z || !x_hat || !k_vectors || !h || !class_weight || !class_bias || !scores_out || !logits_out))
        return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(egnn_outputs_kernel, dim3(node_grid((n_nodes + 3) / 4)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), z,
                       x_hat, k_vectors, n_k, h, class_weight, class_bias, H, num_classes, mask_class, n_nodes, scores_out,
                       logits_out, zero_out, n_zero);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_coord_head(const float* hidden, const float* w_out, const float* coord_diff, const int64_t* offsets,
                        const int64_t* degree, int64_t n_nodes, int H, int spatial_dimension, int mean, float* trans,
                        mdx_stream_t stream)
{
    if (n_nodes < 0 || H < 4 || spatial_dimension < 1) return MDX_ERR_INVALID_ARG;
    if ((H & 3) || spatial_dimension > kWave) return MDX_ERR_UNSUPPORTED;      // one lane per coordinate component
    if (n_nodes == 0) return MDX_OK;
    if (!hidden || !w_out || !coord_diff || !offsets || !degree || !trans) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(egnn_coord_head_kernel, dim3(node_grid(n_nodes)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       hidden, w_out, coord_diff, offsets, degree, n_nodes, H, spatial_dimension, mean, trans);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_segment_rows(const float* data, const int64_t* offsets, const int64_t* degree, int64_t n_nodes, int H, int mean,
                     float* out, mdx_stream_t stream)
{
    if (n_nodes < 0 || H < 4) return MDX_ERR_INVALID_ARG;
    if (H & 3) return MDX_ERR_UNSUPPORTED;
    if (n_nodes == 0) return MDX_OK;
    if (!data || !offsets || !degree || !out) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(segment_rows_kernel, dim3(node_grid(n_nodes)), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream),
                       data, offsets, degree, n_nodes, H, mean, out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

}  // extern "C"
