// Fused EGNN edge chain on the matrix cores: for every edge of the sorted radius graph
//
//     x0 = SiLU(P[src,:H] + P[dst,H:] + b0 + |c_src - c_dst|^2 w_r)                 first message layer (per-node projections P)
//     x  = SiLU(x W_l^T + b_l),  l = 1 .. n_message          -> messages m_e        E_GCL.message_model  (models/egnn.py:136-160)
//     y  = SiLU(y V_l^T + c_l),  l = 1 .. n_coord, y_0 = m                          E_GCL.coord_model    (models/egnn.py:162-200)
//     s_e = y . w_out                                                               last layer Linear(H, 1, bias=False)
//
// in ONE launch: the [edges, H] activations never leave the register file between layers.
//
// Mapping (gfx950, 64-wide wavefronts, one wavefront per SIMD, 512 VGPRs):
//   * a workgroup = 4 wavefronts = a tile of 128 consecutive edges; each wavefront owns 32 edges (columns).
//   * every layer is computed TRANSPOSED, Y^T[n][e] = W[n][:] . X^T[:][e]: the weight matrix is the MFMA A operand, the
//     activations are the B operand.  A 32x32 accumulator tile then holds, per lane, one edge (column = lane & 31) and
//     16 output features (rows 8 (r >> 2) + 4 (lane >> 5) + (r & 3)) -- exactly the lane/element shape of the B operand
//     of the NEXT layer with a permuted k order, so the epilogue (bias + SiLU [+ split]) hands the tile to the next layer
//     in registers: no LDS round trip, no shuffles.  The k permutation is folded into the weight image once, at pack time.
//   * the weights stream through LDS: the image of a layer is cut into chunks of 32 output rows (32 KB at H = 256, the A
//     fragments of one accumulator tile, lane-linear => conflict-free ds_read_b128), copied global -> LDS by direct-to-LDS
//     loads (no VGPRs) into a 3-slot ring, two chunks ahead of the MFMAs, one s_barrier per chunk.  All workgroups stream
//     the same 2.3 MB per E_GCL layer: L2-resident.
//   * two arithmetic modes (same kernel structure, same images' logical content):
//       PREC 0  v_mfma_f32_32x32x2_f32: exact binary32 products and sums (== a k-ordered fmaf chain): the reference's
//               arithmetic up to summation order.  Bound: 157 TFLOP/s.
//       PREC 1  split-f16: x = hi + lo, W = hi + lo (each an f16; hi + lo carries 22 significand bits); the product is
//               hi.hi + (hi.lo + lo.hi) on v_mfma_f32_32x32x16_f16 with binary32 accumulation -- three MFMAs at 16x the
//               f32 rate.  The dropped lo.lo term is 2^-22 relative: the same order as binary32 rounding itself.
//               Magnitudes above the f16 range set MDX_STATUS_EGNN_F16_RANGE (the caller falls back to PREC 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mdx_hip.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_c;
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(1))) const char gbl_c;

constexpr int kWave = 64;
constexpr int kWaves = 4;                 // wavefronts per workgroup = SIMDs per CU
constexpr int kTileEdges = 32 * kWaves;   // edges per workgroup tile
constexpr int kRing = 3;                  // LDS ring slots for weight chunks (two chunks in flight ahead of the MFMAs)
constexpr float kF16Max = 60000.0f;

struct ChainArgs {
    const char* image;          // [layers][H/32 chunks][chunk bytes]
    const float* biases;        // [layers][H]
    const float* bias_in;       // [H]
    const float* w_radial;      // [H]
    const float* w_out;         // [H]
    const float* node_proj;     // [n_nodes][2H]
    const float* coord;         // [n_nodes][D]
    const int64_t* edges;       // [E][2]
    const int64_t* n_edges_dev; // nullable: device-resident edge count (<= n_edges)
    int64_t n_edges;
    int n_message, n_coord, D;
    float* messages;            // [E][H]
    float* edge_scalar;         // [E]
    uint32_t* status;
};

__device__ __forceinline__ float silu_f(float y)
{
    // y * 1 / (1 + exp(-y)): hardware exp2 and reciprocal (~1e-7 relative); -inf / +inf / NaN behave as the formula does
    return y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.44269504088896340736f));
}

// Activations of a wavefront's 32 edges in MFMA B-operand registers.
template <int H, int PREC>
struct Act;
template <int H>
struct Act<H, 0> {
    float v[H / 2];                       // v[16 t + r]: feature 32 t + 8 (r >> 2) + 4 h + (r & 3), h = lane >> 5
};
template <int H>
struct Act<H, 1> {
    half8 hi[H / 16], lo[H / 16];         // k-step s, element j: feature 16 s + 8 (j >> 2) + 4 h + (j & 3)
};

template <int H>
__device__ __forceinline__ void put(Act<H, 0>& a, int t, int r, float y, float& range)
{
    a.v[16 * t + r] = y;
}
template <int H>
__device__ __forceinline__ void put(Act<H, 1>& a, int t, int r, float y, float& range)
{
    const _Float16 hi = (_Float16)y;
    a.hi[2 * t + (r >> 3)][r & 7] = hi;
    a.lo[2 * t + (r >> 3)][r & 7] = (_Float16)(y - (float)hi);
    range = __builtin_fmaxf(range, __builtin_fabsf(y));
}

template <int H, int PREC>
struct Chain {
    static constexpr int NT = H / 32;                 // accumulator tiles (= weight chunks) per layer
    static constexpr int CHUNK = H * 32 * 4;          // bytes: 32 rows x H k x (4 B f32 | 2 B hi + 2 B lo)
    static constexpr int LPW = CHUNK / kWaves / 1024; // direct-to-LDS loads per wavefront per chunk (1 KB each)
    static_assert(LPW >= 1, "chunk smaller than one load per wavefront");

    // ring state (wave-uniform)
    const char* image;
    int chunks_total;       // (n_message + n_coord) * NT
    int next_issue;         // chunk id (mod chunks_total) of the next chunk to request
    int slot_issue;         // ring slot the next request goes to
    int slot_read;          // ring slot of the next chunk to consume
    lds_c* ring;
    int wave, lane;

    __device__ __forceinline__ void issue_chunk()
    {
        const char* src = image + (size_t)next_issue * CHUNK + wave * (CHUNK / kWaves) + lane * 16;
        lds_c* dst = ring + slot_issue * CHUNK + wave * (CHUNK / kWaves);
#pragma unroll
        for (int i = 0; i < LPW; ++i)
            __builtin_amdgcn_global_load_lds((gbl_c*)(src + i * 1024), (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
        next_issue = next_issue + 1 == chunks_total ? 0 : next_issue + 1;
        slot_issue = slot_issue + 1 == kRing ? 0 : slot_issue + 1;
    }

    // Make the next chunk readable, request the one two ahead; returns the LDS address of the readable chunk.
    __device__ __forceinline__ lds_c* acquire_chunk()
    {
        // this wavefront's share of the chunk has landed once at most LPW younger requests (the following chunk) are pending
        if constexpr (LPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (LPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (LPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        // every wavefront's share has landed, and every wavefront has finished reading the slot requested below
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_chunk();
        lds_c* w = ring + slot_read * CHUNK;
        slot_read = slot_read + 1 == kRing ? 0 : slot_read + 1;
        return w;
    }
};

enum LayerKind { kHidden = 0, kLastMessage = 1, kHead = 2 };

template <int H, int PREC>
__global__ __launch_bounds__(kWaves* kWave, 1) void egnn_edge_chain_kernel(ChainArgs p)
{
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    using C = Chain<H, PREC>;
    constexpr int NT = C::NT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    const int h = lane >> 5, col = lane & 31;
    const int layers = p.n_message + p.n_coord;

    lds_c* ring = (lds_c*)lds_raw;
    lds_f* par = (lds_f*)(lds_raw + kRing * C::CHUNK);      // [layers][H] biases | bias_in | w_radial | w_out
    lds_f* par_in = par + layers * H;
    lds_f* par_wr = par_in + H;
    lds_f* par_wo = par_wr + H;

    const int64_t n_edges = p.n_edges_dev ? (*p.n_edges_dev < p.n_edges ? *p.n_edges_dev : p.n_edges) : p.n_edges;
    const int64_t n_tiles = (n_edges + kTileEdges - 1) / kTileEdges;
    if ((int64_t)blockIdx.x >= n_tiles) return;             // uniform per workgroup

    for (int i = threadIdx.x; i < layers * H; i += kWaves * kWave) par[i] = p.biases[i];
    for (int i = threadIdx.x; i < H; i += kWaves * kWave) {
        par_in[i] = p.bias_in[i];
        par_wr[i] = p.w_radial[i];
        par_wo[i] = p.w_out[i];
    }
    __syncthreads();

    C ch;
    ch.image = p.image; ch.chunks_total = layers * NT; ch.next_issue = 0; ch.slot_issue = 0; ch.slot_read = 0;
    ch.ring = ring; ch.wave = wave; ch.lane = lane;
    ch.issue_chunk();
    ch.issue_chunk();

    float range = 0.0f;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // ---- this lane's edge ------------------------------------------------------------------------------------
        const int64_t e_raw = tile * kTileEdges + wave * 32 + col;
        const bool live = e_raw < n_edges;
        const int64_t e = live ? e_raw : n_edges - 1;
        const int64_t src = p.edges[2 * e], dst = p.edges[2 * e + 1];
        float radial = 0.0f;
        for (int k = 0; k < p.D; ++k) {
            const float dlt = p.coord[src * p.D + k] - p.coord[dst * p.D + k];
            radial += dlt * dlt;
        }
        // ---- first message layer, straight into B-operand registers ------------------------------------------------
        Act<H, PREC> xa, xb;
        {
            const float* ps = p.node_proj + src * 2 * H + 4 * h;
            const float* pd = p.node_proj + dst * 2 * H + H + 4 * h;
#pragma unroll
            for (int q = 0; q < H / 8; ++q) {               // features 8 q + 4 h + (0..3)
                const f32x4 a = *(const f32x4*)(ps + 8 * q), b = *(const f32x4*)(pd + 8 * q);
                const f32x4 b0 = *(const __attribute__((address_space(3))) f32x4*)(par_in + 8 * q + 4 * h);
                const f32x4 wr = *(const __attribute__((address_space(3))) f32x4*)(par_wr + 8 * q + 4 * h);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float y = silu_f(((a[i] + b[i]) + b0[i]) + radial * wr[i]);
                    put<H>(xa, q >> 2, 4 * (q & 3) + i, y, range);
                }
            }
        }
        // ---- the chain ----------------------------------------------------------------------------------------------
        float head = 0.0f;
        auto layer = [&](const Act<H, PREC>& in, Act<H, PREC>& out, int l, int kind) {
            const lds_f* bias = par + l * H;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const lds_c* w = ch.acquire_chunk();
                f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                if constexpr (PREC == 0) {
#pragma unroll
                    for (int q = 0; q < H / 8; ++q) {
                        const f32x4 w4 = *(const __attribute__((address_space(3))) f32x4*)(w + q * 1024 + lane * 16);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[i], in.v[4 * q + i], acc, 0, 0, 0);
                    }
                } else {
                    f32x16 cor = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < H / 16; ++s) {
                        const half8 whi = *(const __attribute__((address_space(3))) half8*)(w + s * 2048 + lane * 16);
                        const half8 wlo = *(const __attribute__((address_space(3))) half8*)(w + s * 2048 + 1024 + lane * 16);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, in.hi[s], acc, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(whi, in.lo[s], cor, 0, 0, 0);
                        cor = __builtin_amdgcn_mfma_f32_32x32x16_f16(wlo, in.hi[s], cor, 0, 0, 0);
                    }
                    acc += cor;
                }
                // epilogue of the tile: rows 32 t + 8 g + 4 h + (0..3), g = 0..3
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 b4 = *(const __attribute__((address_space(3))) f32x4*)(bias + 32 * t + 8 * g + 4 * h);
                    f32x4 y;
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = silu_f(acc[4 * g + i] + b4[i]);
                    if (kind == kHead) {
                        const f32x4 wo = *(const __attribute__((address_space(3))) f32x4*)(par_wo + 32 * t + 8 * g + 4 * h);
                        head += (y[0] * wo[0] + y[1] * wo[1]) + (y[2] * wo[2] + y[3] * wo[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i) put<H>(out, t, 4 * g + i, y[i], range);
                        if (kind == kLastMessage && live)
                            *(f32x4*)(p.messages + e * H + 32 * t + 8 * g + 4 * h) = y;
                    }
                }
            }
        };
        // layers alternate between the two register sets; `kind` is wave-uniform
        int l = 0;
        while (l < layers) {
            layer(xa, xb, l, l == p.n_message - 1 ? kLastMessage : (l == layers - 1 ? kHead : kHidden));
            ++l;
            if (l >= layers) break;
            layer(xb, xa, l, l == p.n_message - 1 ? kLastMessage : (l == layers - 1 ? kHead : kHidden));
            ++l;
        }
        head += __shfl_xor(head, 32, kWave);
        if (live && h == 0) p.edge_scalar[e] = head;
    }
    // requests still in flight target this workgroup's LDS: let them land before the workgroup ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (PREC == 1) {
        if (p.status && !(range <= kF16Max)) atomicOr(p.status, MDX_STATUS_EGNN_F16_RANGE);
    }
}

// ---- weight image ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w[MDX_EGNN_CHAIN_MAX_LAYERS];      // [H][H] nn.Linear weights (out, in), device
    int layers, H, precision;
    void* image;
};

__global__ __launch_bounds__(256) void egnn_chain_pack_kernel(PackArgs p)
{
    const int H = p.H, NT = H / 32;
    const int64_t per_layer = (int64_t)H * H;
    const int64_t total = per_layer * p.layers;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per_layer);
        int64_t r = idx - l * per_layer;
        const int t = (int)(r / (32 * H));                  // chunk = accumulator tile
        r -= (int64_t)t * 32 * H;
        (void)NT;
        if (p.precision == 0) {
            // chunk: [q = H/8][lane 64][4 floats]; lane (n = 32 t + (lane & 31), h = lane >> 5) holds W[n][8 q + 4 h + i]
            const int q = (int)(r / 256), lane = (int)(r % 256) / 4, i = (int)(r % 4);
            const int n = 32 * t + (lane & 31), k = 8 * q + 4 * (lane >> 5) + i;
            ((float*)p.image)[idx] = p.w[l][(int64_t)n * H + k];
        } else {
            // chunk: [s = H/16][hi: lane 64 x 8 halfs | lo: lane 64 x 8 halfs]; element j: k = 16 s + 8 (j >> 2) + 4 h + (j & 3)
            const int s = (int)(r / 512), lane = (int)(r % 512) / 8, j = (int)(r % 8);
            const int n = 32 * t + (lane & 31), k = 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
            const float v = p.w[l][(int64_t)n * H + k];
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            _Float16* chunk = (_Float16*)((char*)p.image + ((int64_t)l * (H / 32) + t) * ((int64_t)H * 32 * 4));
            chunk[s * 1024 + lane * 8 + j] = hi;
            chunk[s * 1024 + 512 + lane * 8 + j] = lo;
        }
    }
}

// coord_out[i,:] = coord[i,:] + scale_i sum_{e in segment i} (coord[i,:] - coord[dst_e,:]) * s_e
__global__ __launch_bounds__(256) void egnn_coord_aggregate_kernel(const float* __restrict__ s, const float* __restrict__ coord,
                                                                   const int64_t* __restrict__ edges,
                                                                   const int64_t* __restrict__ offsets,
                                                                   const int64_t* __restrict__ degree, int64_t n_nodes, int D,
                                                                   int mean, float* __restrict__ out)
{
    const int64_t total = n_nodes * D;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t node = idx / D;
        const int k = (int)(idx - node * D);
        const int64_t e0 = offsets[node], deg = degree[node];
        const float ci = coord[idx];
        float acc = 0.0f;
        for (int64_t e = e0; e < e0 + deg; ++e) acc += (ci - coord[edges[2 * e + 1] * D + k]) * s[e];
        if (mean && deg > 0) acc *= 1.0f / (float)deg;
        out[idx] = ci + acc;
    }
}

template <int H, int PREC>
int launch_chain(const ChainArgs& a, int layers, hipStream_t st)
{
    using C = Chain<H, PREC>;
    const size_t lds = (size_t)kRing * C::CHUNK + sizeof(float) * ((size_t)layers * H + 3 * H);
    static bool granted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MDX_ERR_HIP;
    if (lds > 64 * 1024 && !granted[dev]) {
        if (hipFuncSetAttribute((const void*)egnn_edge_chain_kernel<H, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MDX_ERR_HIP;
        granted[dev] = true;
    }
    if (lds > 160 * 1024) return MDX_ERR_UNSUPPORTED;
    int64_t tiles = (a.n_edges + kTileEdges - 1) / kTileEdges;
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);       // persistent: one workgroup per CU
    hipLaunchKernelGGL((egnn_edge_chain_kernel<H, PREC>), dim3(grid), dim3(kWaves * kWave), lds, st, a);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

}  // namespace

extern "C" {

int64_t mdx_egnn_chain_image_bytes(int hidden, int n_layers)
{
    if (hidden < 32 || (hidden % 32) || n_layers < 1) return -1;
    return (int64_t)n_layers * hidden * hidden * 4;
}

int mdx_egnn_chain_pack(const float* const* weights_host, int n_layers, int hidden, int precision, void* image_out,
                        mdx_stream_t stream)
{
    if (!weights_host || !image_out || n_layers < 1 || (precision != 0 && precision != 1)) return MDX_ERR_INVALID_ARG;
    if (n_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (hidden != 32 && hidden != 64 && hidden != 128 && hidden != 256) return MDX_ERR_UNSUPPORTED;
    PackArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        if (!weights_host[l]) return MDX_ERR_INVALID_ARG;
        a.w[l] = weights_host[l];
    }
    a.layers = n_layers; a.H = hidden; a.precision = precision; a.image = image_out;
    const int64_t total = (int64_t)n_layers * hidden * hidden;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(egnn_chain_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), a);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_edge_chain(const mdx_egnn_chain_t* c, const float* node_proj, const float* coord, int coord_dimension,
                        const int64_t* edges, int64_t n_edges, const int64_t* n_edges_dev, float* messages_out,
                        float* edge_scalar_out, uint32_t* status, mdx_stream_t stream)
{
    if (!c || n_edges < 0 || coord_dimension < 1) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers < 1 || c->n_coord_layers < 1 || (c->precision != 0 && c->precision != 1))
        return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers + c->n_coord_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (c->hidden != 32 && c->hidden != 64 && c->hidden != 128 && c->hidden != 256) return MDX_ERR_UNSUPPORTED;
    if (n_edges == 0) return MDX_OK;
    if (!c->weight_image || !c->biases || !c->bias_in || !c->w_radial || !c->w_out || !node_proj || !coord || !edges ||
        !messages_out || !edge_scalar_out)
        return MDX_ERR_INVALID_ARG;
    ChainArgs a{};
    a.image = (const char*)c->weight_image; a.biases = c->biases; a.bias_in = c->bias_in; a.w_radial = c->w_radial;
    a.w_out = c->w_out; a.node_proj = node_proj; a.coord = coord; a.edges = edges; a.n_edges_dev = n_edges_dev;
    a.n_edges = n_edges; a.n_message = c->n_message_layers; a.n_coord = c->n_coord_layers; a.D = coord_dimension;
    a.messages = messages_out; a.edge_scalar = edge_scalar_out; a.status = status;
    const int layers = a.n_message + a.n_coord;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MDX_CHAIN_CASE(HH)                                                                         \
    case HH: return c->precision == 0 ? launch_chain<HH, 0>(a, layers, st) : launch_chain<HH, 1>(a, layers, st);
    switch (c->hidden) {
        MDX_CHAIN_CASE(32)
        MDX_CHAIN_CASE(64)
        MDX_CHAIN_CASE(128)
        MDX_CHAIN_CASE(256)
    }
#undef MDX_CHAIN_CASE
    return MDX_ERR_UNSUPPORTED;
}

int mdx_egnn_coord_aggregate(const float* edge_scalar, const float* coord, int coord_dimension, const int64_t* edges,
                             const int64_t* offsets, const int64_t* degree, int64_t n_nodes, int mean, float* coord_out,
                             mdx_stream_t stream)
{
    if (n_nodes < 0 || coord_dimension < 1) return MDX_ERR_INVALID_ARG;
    if (n_nodes == 0) return MDX_OK;
    if (!edge_scalar || !coord || !edges || !offsets || !degree || !coord_out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_nodes * coord_dimension + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_coord_aggregate_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       edge_scalar, coord, edges, offsets, degree, n_nodes, coord_dimension, mean, coord_out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

}  // extern "C"
