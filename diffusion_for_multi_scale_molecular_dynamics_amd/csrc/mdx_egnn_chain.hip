// Fused EGNN edge chain on the matrix cores: for every edge of the sorted radius graph
//
//     x0 = SiLU(P[src,:H] + P[dst,H:] + b0 + |c_src - c_dst|^2 w_r)                 first message layer (per-node projections P)
//     x  = SiLU(x W_l^T + b_l),  l = 1 .. n_message          -> messages m_e        E_GCL.message_model  (models/egnn.py:136-160)
//     y  = SiLU(y V_l^T + c_l),  l = 1 .. n_coord, y_0 = m                          E_GCL.coord_model    (models/egnn.py:162-200)
//     s_e = y . w_out                                                               last layer Linear(H, 1, bias=False)
//
// in ONE launch: the [edges, H] activations never leave the register file between layers.  With the layer's `attention` option
// (models/egnn.py:148-160: m_e <- m_e sigmoid(m_e . w_att + b_att)) the gate is applied where the messages are complete --
// between the last message layer and the first coordinate layer, on the operand registers -- so both the message sums and the
// coordinate MLP see the gated messages (instantiations ATT = true; the others are compiled without a trace of it).
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
//     loads (no VGPRs) into a 4-slot ring, three chunks ahead of the MFMAs, one s_barrier per chunk in the middle of a
//     tile; a wavefront requests one quarter of each chunk, one 1-KB piece per two k-steps.  All workgroups stream the
//     same 2.4 MB per E_GCL layer: L2-resident.
//   * tiles are dealt to workgroups XCD by XCD (each XCD one contiguous eighth of the edge list: its L2 holds the node rows of
//     two or three structures beside the weight image); with message_mode = piece sums the messages are added up per node
//     inside the kernel (a segmented scan over DPP rows) and only those sums are written.
//   * two arithmetic modes (same kernel structure, same images' logical content):
//       PREC 0  v_mfma_f32_32x32x2_f32: exact binary32 products and sums (== a k-ordered fmaf chain): the reference's
//               arithmetic up to summation order.  Bound: 157 TFLOP/s.
//       PREC 1  split-f16: x = hi + lo, W = hi + lo (each an f16; hi + lo carries 22 significand bits); the product is
//               hi.hi + (hi.lo + lo.hi) on v_mfma_f32_32x32x16_f16 with binary32 accumulation -- three MFMAs at 16x the
//               f32 rate.  The dropped lo.lo term is 2^-22 relative: the same order as binary32 rounding itself.
//               f16 has 5 exponent bits: unscaled, lo is a subnormal (quantum 2^-24) as soon as |v| < 2^-3, and the split no
//               longer carries 22 bits.  So both operands are held SCALED BY EXACT POWERS OF TWO: the image of layer l is
//               2^a_l W_l with a_l chosen at pack time so that the largest |W| sits in [2^13, 2^14) (every element down to
//               2^-16 of the largest keeps its 22 bits), and the carried activations are 2^b u, b = kActExp (22 bits down to
//               |u| = 2^-2-b; overflow beyond 65504 / 2^b).  The accumulator then holds 2^(a_l+b) z; the epilogue's first
//               instruction brings the exponent back (Scale below).  Scaling by a power of two commutes with every
//               rounding involved, so data of order one gives the same bits as without it.
//               Magnitudes above the f16 range set MDX_STATUS_EGNN_F16_RANGE (the caller falls back to PREC 0).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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
constexpr int kFastD = 6;                 // coordinate components per node with an unrolled tile top (the EGNN in three dimensions)
constexpr int kRing = 4;                  // LDS ring slots for weight chunks: being read | readable next | two in flight
constexpr int kScaleSlots = MDX_EGNN_CHAIN_MAX_LAYERS + 4;   // struct Scale per packed layer (+ the head / two projection layers) + the
                                                             // four factors at the chain's ends (kEnds below)
constexpr int kMaxPositions = MDX_EGNN_CHAIN_MAX_LAYERS + 2; // places where a chain carries activations (see kActExp)
constexpr int kEnds = kScaleSlots - 1;                       // slot of {2^-c0, log2(e) 2^c0, ln 2 2^-c_out, log2(e) 2^c_reread}

struct ChainArgs {
    const char* image;          // [layers][H/32 chunks][chunk bytes]
    const float* biases;        // [layers][H]
    const float* bias_in;       // [H]
    const float* w_radial;      // [H]
    const int32_t* exps;        // [packed layers + 1] split-f16: the image holds 2^exps[l] W_l (last: the head row); else null
    const int32_t* act_exps;    // [layers + 2] split-f16, nullable: position q carries 2^act_exps[q] u (null: kActExp everywhere)
    uint32_t* act_max;          // [layers + 2] exact-f32 kernels, nullable: running maximum (float bits) of |u| per position
    const float* node_proj;     // [n_nodes][2H]
    const float* coord;         // [n_nodes][D]
    const int64_t* edges;       // [E][2]
    const int64_t* n_edges_dev; // nullable: device-resident edge count (<= n_edges)
    int64_t n_edges;
    int n_message, n_coord, D;
    int piece_sums;             // the caller asked for MODE 2: `messages` receives per-node PIECE SUMS (see aggregate_pieces)
    float* messages;            // [E][H]; piece sums: [ceil(E / 16) + n_nodes][H]
    float* edge_scalar;         // [E]
    uint32_t* status;
    // MODE 1 (a chain of H -> H layers over the ROWS of a matrix, last layer linear, optional residual): n_edges rows
    const float* rows_in;       // [M][ld_in]: the first H columns (MODE 3: [M][2H] = h | agg, ld_in = 2 H)
    const float* residual;      // [M][ld_in] first H columns, nullable
    int64_t ld_in;              // row stride of rows_in and residual, in floats
    const float* rows_in2;      // MODE 3: the second H columns (agg) as rows of their own, row stride ld_in2
    int64_t ld_in2;
    const float* att_w;         // ATT instantiations: [H] weight of E_GCL.att_mlp's Linear(H, 1); att_b: [1] its bias
    const float* att_b;
    float* proj_out;            // MODE 3, nullable: [M][2H] = out W_p^T for the 2 H x H layers that follow the MLP in the image
    float* rows_out;            // [M][H]
    void* stamp_buf;            // -DMDX_CHAIN_STAMPS builds only (else null): see MDX_STAMP_WRITE
};

constexpr float kLog2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;

// Inside the chain every activation is carried as u = log2(e) SiLU(y), computed from the pre-activation z = log2(e) y:
//   u = z / (1 + 2^-z)      -- hardware exp2 (negation is a source modifier) and reciprocal, four instructions per value.
// W u = log2(e) W SiLU(y), so a layer fed with u and started from log2(e) b produces the next z directly: only the biases
// (scaled when they are staged into LDS), the first layer (scaled here) and the two outputs (messages x ln 2 when they are
// stored; w_out x ln 2 in the packed image) know about the factor.
__device__ __forceinline__ float silu_scaled(float z)
{
    return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z));
}

// Split-f16: the carried activations are 2^c u (see the header comment), c = kActExp unless the caller hands over exponents
// of its own.  2^6: full 22 bits for |u| >= 2^-8, an absolute floor of 2^-31 below that; the f16 range is left at |u| > 1023
// (|SiLU| > 709), which the status word reports.
// POSITIONS.  A chain of L layers has L + 2 places where activations are carried: position q (0 <= q <= L) = the operand
// entering layer q (q = 0: the first layer's output of the edge chain / the rows of a row chain; q = L: what leaves the last
// layer: the head's operand / the rows written out), position L + 1 = the rows read back in front of the projection layers
// (MODE 3).  ChainArgs::act_exps gives one exponent per position (an activation beyond 65504 / 2^c sets the range bit), and
// ChainArgs::act_max collects, in the exact-f32 kernels, the largest |u| seen at each position -- what the caller derives the
// exponents from (mdx_egnn_chain_adapt_activation_exponents).  MODE 3: positions 0 and 1 are one (h and agg feed the two
// halves of ONE layer whose accumulators continue one another): position 0 is used for both.
template <int PREC>
constexpr int kActExp = PREC >= 1 ? 6 : 0;
constexpr float pow2_const(int e) { float v = 1.0f; for (int i = 0; i < (e < 0 ? -e : e); ++i) v = e < 0 ? v * 0.5f : v * 2.0f; return v; }
__device__ __forceinline__ float pow2_bits(int e) { return __builtin_bit_cast(float, (uint32_t)(127 + e) << 23); }   // |e| <= 126

// Exponent bookkeeping of one layer (wave-uniform; staged in LDS at kernel start, carried in scalar registers):
// the accumulator of a tile of layer l holds A = 2^(a+b) z (a = the image's exponent, b = the exponent of the layer's operand,
// b' = that of its output: the next position).
//   neg_c = -2^-(a+b)     t = A neg_c = -z                       (the one instruction the scaling costs per value)
//   k     =  2^(a+b-b')   u' = A / (k + k 2^t) = 2^b' z / (1 + 2^-z)
//   inv_a =  2^-(a+b-b')  a layer without activation: u' = A inv_a = 2^b' z
//   out   =  ln 2 2^-(a+b)   a tile stored as it is (the head, the projections behind the node MLP): y = A out
struct Scale {
    float neg_c, k, inv_a, out;
};
// u' = 2^b u for the first layer, computed on the vector ALU from z itself; kb = 2^-b
template <int PREC>
__device__ __forceinline__ float silu_first(float z, float kb)
{
    if constexpr (PREC == 0) return silu_scaled(z);
    return z * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(-z), kb, kb));
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
// PREC 2: split-f16 on v_mfma_f32_16x16x32_f16.  A lane holds TWO edges (column groups cg = 0, 1: edges 16 cg + (lane & 15) of
// the wavefront's 32) and a quarter of their features (g = lane >> 4).  A 32-row chunk is two row tiles rho of 16; its sixteen
// accumulator registers are r = 8 rho + 4 cg + i  <->  feature 32 t + 16 rho + 4 g + i of edge cg: exactly element j = 4 rho + i
// of the B fragment of k-step t (32 features deep) of column group cg, so the hand-over to the next layer stays in registers.
template <int H>
struct Act<H, 2> {
    half8 hi[2][H / 32], lo[2][H / 32];   // [cg][k-step s], element j: feature 32 s + 16 (j >> 2) + 4 g + (j & 3)
};

// Which of the lane's edges a float4 group q (registers 4 q .. 4 q + 3 of a tile) belongs to, and its first feature within the
// 32-row chunk (sub = lane >> 5 for the 32x32 shapes, lane >> 4 for the 16x16 shape).
template <int PREC>
struct Lay {
    static constexpr bool W16 = PREC == 2;
    static constexpr int NS = W16 ? 2 : 1;                   // edges (or rows) per lane
    static __device__ __forceinline__ constexpr int es(int q) { return W16 ? (q & 1) : 0; }
    static __device__ __forceinline__ int fb(int q, int sub) { return W16 ? 16 * (q >> 1) + 4 * sub : 8 * q + 4 * sub; }
};

template <int H>
__device__ __forceinline__ void put(Act<H, 0>& a, int t, int r, float y)
{
    a.v[16 * t + r] = y;
}
template <int H>
__device__ __forceinline__ void put(Act<H, 1>& a, int t, int r, float y)
{
    const _Float16 hi = (_Float16)y;
    a.hi[2 * t + (r >> 3)][r & 7] = hi;
    a.lo[2 * t + (r >> 3)][r & 7] = (_Float16)(y - (float)hi);
}

// Elements r, r + 1 (r even) of tile t.  Split-f16: hi = f16(y) (packed convert), lo = f16(y - hi) with the subtraction
// straight off the packed halves (v_fma_mix_f32: f16 operand x -1 + f32 operand, exact) -- four instructions per pair.
template <int H>
__device__ __forceinline__ void put_pair(Act<H, 0>& a, int t, int r, float y0, float y1)
{
    a.v[16 * t + r] = y0;
    a.v[16 * t + r + 1] = y1;
}
__device__ __forceinline__ void split_f16_pair(float y0, float y1, uint32_t& hi, uint32_t& lo)
{
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(y0), "v"(y1));
    float l0, l1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(hi), "v"(y0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(hi), "v"(y1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(l0), "v"(l1));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int H>
__device__ __forceinline__ void put_pair(Act<H, 1>& a, int t, int r, float y0, float y1)
{
    uint32_t hi, lo;
    split_f16_pair(y0, y1, hi, lo);
    const int s = 2 * t + (r >> 3), j = (r & 7) >> 1;
    u32x4 vh = __builtin_bit_cast(u32x4, a.hi[s]), vl = __builtin_bit_cast(u32x4, a.lo[s]);
    vh[j] = hi;
    vl[j] = lo;
    a.hi[s] = __builtin_bit_cast(half8, vh);
    a.lo[s] = __builtin_bit_cast(half8, vl);
}
template <int H>
__device__ __forceinline__ void put_pair(Act<H, 2>& a, int t, int r, float y0, float y1)
{
    uint32_t hi, lo;
    split_f16_pair(y0, y1, hi, lo);
    const int cg = (r >> 2) & 1, j = 2 * (r >> 3) + ((r & 3) >> 1);      // register r = 8 rho + 4 cg + i  ->  element 4 rho + i
    u32x4 vh = __builtin_bit_cast(u32x4, a.hi[cg][t]), vl = __builtin_bit_cast(u32x4, a.lo[cg][t]);
    vh[j] = hi;
    vl[j] = lo;
    a.hi[cg][t] = __builtin_bit_cast(half8, vh);
    a.lo[cg][t] = __builtin_bit_cast(half8, vl);
}
template <int H>
__device__ __forceinline__ void put(Act<H, 2>& a, int t, int r, float y)
{
    const _Float16 hi = (_Float16)y;
    const int cg = (r >> 2) & 1, j = 4 * (r >> 3) + (r & 3);
    a.hi[cg][t][j] = hi;
    a.lo[cg][t][j] = (_Float16)(y - (float)hi);
}

// The four carried values of float4 group q of tile t (registers 4 q .. 4 q + 3), as binary32: hi + lo in one instruction per
// value for the split forms (v_fma_mix_f32 with both addends taken from the packed halves).
template <int H>
__device__ __forceinline__ f32x4 get4(const Act<H, 0>& a, int t, int q)
{
    return f32x4{a.v[16 * t + 4 * q], a.v[16 * t + 4 * q + 1], a.v[16 * t + 4 * q + 2], a.v[16 * t + 4 * q + 3]};
}
__device__ __forceinline__ void join_f16_pair(uint32_t ph, uint32_t pl, float& y0, float& y1)
{
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(y0) : "v"(ph), "v"(pl));
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(y1) : "v"(ph), "v"(pl));
}
template <int H>
__device__ __forceinline__ f32x4 get4(const Act<H, 1>& a, int t, int q)
{
    const int r0 = 4 * q;
    const u32x4 vh = __builtin_bit_cast(u32x4, a.hi[2 * t + (r0 >> 3)]), vl = __builtin_bit_cast(u32x4, a.lo[2 * t + (r0 >> 3)]);
    f32x4 y;
    float y0, y1;
    join_f16_pair(vh[(r0 & 7) >> 1], vl[(r0 & 7) >> 1], y0, y1);
    y[0] = y0; y[1] = y1;
    join_f16_pair(vh[((r0 & 7) >> 1) + 1], vl[((r0 & 7) >> 1) + 1], y0, y1);
    y[2] = y0; y[3] = y1;
    return y;
}
template <int H>
__device__ __forceinline__ f32x4 get4(const Act<H, 2>& a, int t, int q)
{
    const int cg = q & 1, rho = q >> 1;                                   // registers 8 rho + 4 cg + (0..3): elements 4 rho + (0..3)
    const u32x4 vh = __builtin_bit_cast(u32x4, a.hi[cg][t]), vl = __builtin_bit_cast(u32x4, a.lo[cg][t]);
    f32x4 y;
    float y0, y1;
    join_f16_pair(vh[2 * rho], vl[2 * rho], y0, y1);
    y[0] = y0; y[1] = y1;
    join_f16_pair(vh[2 * rho + 1], vl[2 * rho + 1], y0, y1);
    y[2] = y0; y[3] = y1;
    return y;
}

// Timing diagnostics (never in the shipped build): -DMDX_CHAIN_STAMPS makes wavefront 0 of workgroup 0 write s_memtime at
// marked points into the buffer passed as `status` (uint64 [4096]); tools/chain_bench.py --stamps prints the intervals.
// -DMDX_CHAIN_STAMPS=2: only the two ends of the kernel, with s_memtime (shader clock) AND s_memrealtime (100 MHz): the
// clock the chip held during the launch (tools/chain_bench.py --clocks).
// The stamp list is the caller's buffer (uint64 [4096], passed where the status word goes): entries [0, 4000), the entry
// count in element 4095 (zeroed on the launch stream by the entry point).  `stamp_buf_` must be in scope: the kernel's
// copy of ChainArgs::stamp_buf, or Chain's member.  (No __device__ globals, no host-side symbol copies: the first version
// set two globals with hipMemcpyToSymbolAsync from stack variables -- read by the copy after the entry point had
// returned; those builds died at process exit.)
#ifdef MDX_CHAIN_STAMPS
#define MDX_STAMP_WRITE(id, clock)                                                                                 \
    do {                                                                                                           \
        unsigned long long* sb_ = (unsigned long long*)stamp_buf_;                                                 \
        if (sb_ && blockIdx.x == 0 && threadIdx.x == 0 && sb_[4095] < 4000) {                                      \
            sb_[sb_[4095]] = ((unsigned long long)(id) << 48) | ((clock) & 0xffffffffffffull);                     \
            sb_[4095] += 1;                                                                                        \
        }                                                                                                          \
    } while (0)
#define MDX_STAMP_ALWAYS(id) MDX_STAMP_WRITE(id, __builtin_amdgcn_s_memtime())
#define MDX_STAMP_REALTIME(id) MDX_STAMP_WRITE(id, __builtin_amdgcn_s_memrealtime())
#if MDX_CHAIN_STAMPS >= 2
#define MDX_STAMP(id)
#else
#define MDX_STAMP(id) MDX_STAMP_ALWAYS(id)
#endif
#else
#define MDX_STAMP(id)
#define MDX_STAMP_ALWAYS(id)
#define MDX_STAMP_REALTIME(id)
#endif

template <int H, int PREC, bool GUARD = true>
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

    int share;              // byte offset, inside every chunk, of the quarter this wavefront requests: (wave + workgroup's
                            // rotation) mod 4 quarters (all workgroups stream the same image in the same order at nearly
                            // the same time: without the rotation every CU of an XCD asks its L2 for the same lines at once)
    uint32_t lane16;        // lane * 16: the per-lane part of every request's address

    // The chunk being requested, piece by piece: LPW direct-to-LDS loads per wavefront (1 KB each: its quarter of the
    // chunk), ONE per STEPS / LPW k-steps, each in the issue shadow of an MFMA.  (All LPW at once, right behind the
    // barrier, is 32 KB through the CU's one address path: every wavefront sat ~370 cycles per tile in the issue of its
    // eight loads, with its matrix pipe idle.)  Addresses are scalar: begin_chunk() forms the quarter's global base and LDS
    // base once; a piece's kilobytes are its instruction offset.
    int issue_id, issue_slot;
#ifdef MDX_CHAIN_VERIFY
    uint32_t* verify_status = nullptr;
#endif
#ifdef MDX_CHAIN_STAMPS
    void* stamp_buf_ = nullptr;
#endif
    uint32_t issue_src;     // byte offset of (chunk, share) in the image   (SGPR; the image is a few megabytes)
    uint32_t issue_dst;     // LDS byte address of slot + share           (SGPR)

    // A request is ordered against the compiler's own memory instructions ("memory"): the counted vmcnt waits of acquire_next
    // assume the message / piece-row stores and the requests are issued in program order.  (m0 is written and read inside
    // one statement: the compiler reserves it and keeps nothing in it across statements; naming it as a clobber only draws
    // "clobber list contains reserved registers".)
    __device__ __forceinline__ void issue_piece(int i)
    {
        // Written as assembly ON PURPOSE.  The compiler's wait-count pass files the builtin (a FLAT-encoded instruction
        // with an LDS operand) as an access that may complete out of order on BOTH counters; with one always in flight,
        // every wait for a weight fragment became `s_waitcnt lgkmcnt(0)` -- the fragments just requested for the steps
        // ahead included.  The requests are waited for by hand anyway (acquire_next); the compiler's own vmcnt waits stay
        // correct with instructions it does not see in flight: they can only wait for more than they need.
        // (m0 is a reserved register: the compiler never keeps a value in it across statements.)
        // Scalar base (image + the offset begin_chunk() left in a scalar register, + 4 KB for pieces 4 .. 7) as the SADDR
        // operand, lane * 16 as the 32-bit vector offset, the piece's kilobytes as the instruction offset -- which applies
        // to BOTH the global and the LDS address (tools/dma_probe.hip): no vector instruction per request.
        const uint64_t src = (uint64_t)(uintptr_t)image + (uint64_t)(issue_src + (uint32_t)((i >> 2) * 4096));
        const uint32_t dst = issue_dst + (uint32_t)((i >> 2) * 4096);
#ifdef MDX_CHAIN_VERIFY
        {
            // debugging aid: the addresses a request is about to use against the plain formula; a mismatch is reported in the
            // status word (bit 30 source, bit 31 destination) and the plain values are used
            const uint64_t ref_src = (uint64_t)(uintptr_t)image + (uint64_t)issue_id * CHUNK + (uint64_t)share + (uint64_t)(i * 1024) + lane16;
            const uint32_t ref_dst = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring + (uint32_t)(issue_slot * CHUNK + share + i * 1024));
            if (verify_status && src + lane16 + (i & 3) * 1024 != ref_src) atomicOr(verify_status, 1u << 30);
            if (verify_status && dst + (i & 3) * 1024 != ref_dst) atomicOr(verify_status, 1u << 31);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(ref_src), "s"(ref_dst));
        }
#else
        if constexpr (SPREAD) {
            // (s_nop 3: with the s_mov, five wait states between whatever precedes this statement and the request.  The
            // compiler may restore a spilled scalar register with v_readlane right before it, and a scalar register written
            // by a vector instruction is not yet readable as a memory instruction's address for five wait states -- a rule
            // it enforces for its own instructions only.)
            if constexpr (GUARD)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(lane16), "s"(src), "s"(dst),
                             "n"((i & 3) * 1024) : "memory");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" ::"v"(lane16), "s"(src), "s"(dst),
                             "n"((i & 3) * 1024) : "memory");
        } else {
            // (the burst form of the exact-f32 chain: the per-lane address as a vector-register pair -- with eight scalar-base
            // requests in a row the row-chain instantiation at H = 256 spills 3.6 KB per lane)
            const uint64_t lane_src = src + lane16;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2" ::"v"(lane_src), "s"(dst), "n"((i & 3) * 1024) : "memory");
        }
#endif
    }
    // the next chunk of the stream becomes the one being requested
    __device__ __forceinline__ void begin_chunk()
    {
        issue_id = next_issue;
        issue_slot = slot_issue;
        scalar_addresses();
        next_issue = next_issue + 1 == chunks_total ? 0 : next_issue + 1;
        slot_issue = slot_issue + 1 == kRing ? 0 : slot_issue + 1;
    }
    // issue_src / issue_dst of the chunk being requested, in scalar registers.  v_readfirstlane as assembly: the builtin is
    // folded away on a value the compiler knows to be uniform, which then stays in the vector registers it has chosen for
    // the ring state -- and an "s" operand of the request would be handed a vector register.  Also called at the top of
    // every edge tile: a value carried around the tile loop may be moved to vector registers again.
    __device__ __forceinline__ void scalar_addresses()
    {
        const uint32_t src = (uint32_t)issue_id * (uint32_t)CHUNK + (uint32_t)share;
        const uint32_t dst = (uint32_t)(uintptr_t)ring + (uint32_t)(issue_slot * CHUNK + share);
        if constexpr (!SPREAD) {
            // the burst form (exact-f32 chain) uses them at once, the source as part of a vector address: no assembly needed
            issue_src = src;
            issue_dst = __builtin_amdgcn_readfirstlane(dst);
            return;
        }
        uint32_t s_src, s_dst;
        // (s_nop 1 first: a vector register written by the instruction just before is not yet readable by v_readfirstlane --
        // a wait state the compiler inserts for its own instructions and cannot see into this statement; without it the
        // first of the two came back wrong now and then.  s_nop 4 last: a scalar register written by a vector
        // instruction must not be read as a memory instruction's address within five wait states.)
        asm volatile("s_nop 1\n\tv_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3\n\ts_nop 4" : "=s"(s_src), "=s"(s_dst) : "v"(src), "v"(dst));
        issue_src = s_src;
        issue_dst = s_dst;
    }
    // pieces requested in the SAME tile as the chunk's begin_chunk() (behind the acquire in the middle of the tile); the
    // others follow in the first half of the next tile
    // (split-f16 only: the exact-f32 chain, whose MFMAs are twice as long, measured 2 % faster with the burst)
    static constexpr bool SPREAD = PREC >= 1;
    static constexpr int FIRST_HALF = !SPREAD ? LPW : (LPW >= 2 ? LPW / 2 : 1);

    // Called in the MIDDLE of a tile's MFMA stream: makes the NEXT chunk readable (so that its first fragments can be read
    // beside the second half of the current tile's MFMAs) and opens the requests of the chunk three ahead into the slot of
    // the chunk before the current one -- every wavefront is past that one.  Returns the LDS address of the next chunk.
    int stores_count;       // store instructions issued after the last chunk request (messages, piece sums, rows): they are
                            // younger than the chunk waited for next, so that wait may leave them pending too; any LOWER
                            // bound of the true number is safe

    __device__ __forceinline__ lds_c* acquire_next()
    {
        // this wavefront's share of the next chunk has landed once at most LPW younger requests (the chunk after it) are
        // pending; the barrier extends that to every wavefront's share
        MDX_STAMP(1);
        if (stores_count > 0) {
            // leave up to LPW requests + the youngest stores pending (any lower bound of the store count is safe)
            const int k = stores_count >= 48 ? 48 : (stores_count >= 32 ? 32 : (stores_count >= 16 ? 16 : (stores_count >= 8 ? 8 : 0)));
            stores_count = 0;
            if constexpr (LPW == 8) {
                if (k == 48) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
                else if (k == 32) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
                else if (k == 16) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (k == 8) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if constexpr (LPW == 4) {
                if (k == 48) asm volatile("s_waitcnt vmcnt(52)" ::: "memory");
                else if (k == 32) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                else if (k == 16) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                else if (k == 8) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else if constexpr (LPW == 2) {
                if (k >= 8) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else {
                if (k >= 8) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            }
        } else {
            if constexpr (LPW == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (LPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (LPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
        MDX_STAMP(2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        MDX_STAMP(3);
        begin_chunk();
        if constexpr (!SPREAD) {
#pragma unroll
            for (int i = 0; i < LPW; ++i) issue_piece(i);
        }
        slot_read = slot_read + 1 == kRing ? 0 : slot_read + 1;
        return ring + slot_read * CHUNK;
    }

    // Once per workgroup, before the first tile: two chunks and the first half of the third requested, the first one readable.
    __device__ __forceinline__ lds_c* prime()
    {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            begin_chunk();
#pragma unroll
            for (int i = 0; i < LPW; ++i)
                if (c < 2 || i < FIRST_HALF) issue_piece(i);
        }
        // the first chunk has landed once at most the LPW + FIRST_HALF younger requests are pending
        constexpr int YOUNGER = LPW + FIRST_HALF;
        if constexpr (YOUNGER == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (YOUNGER == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (YOUNGER == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (YOUNGER == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (YOUNGER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (YOUNGER == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else { static_assert(YOUNGER == 2, "unexpected chunk geometry"); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        return ring + slot_read * CHUNK;
    }
};

// The outstanding epilogue: it runs one tile late, BESIDE the next tile's MFMAs (the matrix pipe and the vector ALU issue
// from one wavefront's stream; with one wavefront per SIMD nothing else would fill the MFMAs' issue shadow).
// Elements [r0, r1) of the pending tile tp: u = z / (1 + 2^-z), z = the accumulator (the scaled bias is already in it: it was
// its initial value) -> the next layer's operand registers.  A value beyond the f16 range becomes an infinity in the split
// and a NaN one layer later, in every feature of its edge: it reaches the kernel's outputs, where it is looked for.  Branch-free, so that it is one scheduling region with the MFMAs around
// it.  tp, r0, r1 are constants after unrolling.
// `linear` (wave-uniform): the tile belongs to a layer without activation (the last layer of a row chain): u = z, a select.
// `amax` (exact-f32 kernels only): running maximum of |u| over the values written (ChainArgs::act_max).
template <int H, int PREC>
__device__ __forceinline__ void epilogue_elements(int tp, int r0, int r1, const f32x16& pend, Act<H, PREC>& dst,
                                                  const Scale& sc, bool linear, float& amax)
{
    // split-f16: the accumulator is 2^(a+b) z (struct Scale): -z, then 2^b z / (1 + 2^-z) with the divisor pre-scaled
    auto act = [&](float A) -> float {
        if constexpr (PREC == 0) return silu_scaled(A);
        else return A * __builtin_amdgcn_rcpf(__builtin_fmaf(__builtin_amdgcn_exp2f(A * sc.neg_c), sc.k, sc.k));
    };
    auto lin = [&](float A) -> float { return PREC == 0 ? A : A * sc.inv_a; };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (r < r0 || r >= r1) continue;
        if constexpr (PREC == 0) {
            // (one value at a time: in pairs this instantiation nearly doubles its run time -- 7.25 -> 12.9 ms at H = 256)
            const float y = linear ? lin(pend[r]) : act(pend[r]);
            amax = __builtin_fmaxf(amax, __builtin_fabsf(y));
            put<H>(dst, tp, r, y);
        } else if (!(r & 1)) {
            const float y0 = linear ? lin(pend[r]) : act(pend[r]), y1 = linear ? lin(pend[r + 1]) : act(pend[r + 1]);
            put_pair<H>(dst, tp, r, y0, y1);
        }
    }
}

// The same epilogue as a three-stage software pipeline over the k-steps of the tile beside which it runs (split-f16, H = 256:
// 16 k-steps, 8 pairs of values).  Pair j enters at step kPairStart[j]:
//   stage A (that step)   A from the accumulator, t = -z, e = 2^t          2 (moves) + 2 + 2 instructions
//   stage B (the next)    y = A / (k + k e)                                2 + 2 + 2
//   stage C (the one after) split y into f16 halves -> operand registers   3 (4)
// so every step carries about ten vector instructions beside its three MFMAs, instead of none beside the first half of the
// tile's MFMAs and twenty beside each step of the second half -- which is what the scheduler makes of the plain form
// (tools/tile_stats.py) -- and no step waits on a chain of nine dependent instructions.  The last pair is complete at the
// end of step 13: the tile that follows a layer's last tile reads these values in its steps 14 and 15 (the 16x16 shape: all
// sixteen as one k-step of 32 in both; the 32x32 shape: eight in each).
// The six operations of a value -- 0: A from the accumulator, t = A c   1: e = 2^t   2: d = k + k e   3: r = 1 / d   4: y = A r
// 5: split into the operand registers -- are dealt to the steps by kOpStep (step offset of each operation from the pair's
// first step).
constexpr int kPairStart[8] = {0, 1, 3, 4, 6, 8, 9, 11};
constexpr int kOpStep[6] = {0, 0, 1, 1, 1, 2};
template <int H>
struct EpiloguePipe {
    float a[8][2], e[8][2];
    template <int PREC>
    __device__ __forceinline__ void step(int s, int tp, const f32x16& pend, Act<H, PREC>& dst, const Scale& sc, bool linear)
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int op = 0; op < 6; ++op) {
                if (s != kPairStart[j] + kOpStep[op]) continue;
                if (op == 5) {
                    put_pair<H>(dst, tp, 2 * j, e[j][0], e[j][1]);
                    continue;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    if (op == 0) {
                        a[j][i] = pend[2 * j + i];
                        e[j][i] = a[j][i] * sc.neg_c;
                    } else if (op == 1) {
                        e[j][i] = __builtin_amdgcn_exp2f(e[j][i]);
                    } else if (op == 2) {
                        e[j][i] = __builtin_fmaf(e[j][i], sc.k, sc.k);
                    } else if (op == 3) {
                        e[j][i] = __builtin_amdgcn_rcpf(e[j][i]);
                    } else {
                        e[j][i] = linear ? a[j][i] * sc.inv_a : a[j][i] * e[j][i];
                    }
                }
            }
        }
    }
};

// One step of the segmented scan over a DPP row of 16 lanes, for the sixteen values of a tile slice: x[r] += (the value D lanes
// down the row; 0 beyond the row's first lane) * gate, gate = 1.0 where that lane belongs to this lane's piece, else 0.0.
// ONE instruction per value: v_fmac_f32 with the DPP row shift on its first source (bound_ctrl: a lane without a source in its
// row reads 0).  Written as ONE assembly block per step: the compiler selects `update_dpp` + `fmaf` as v_mov_b32_dpp + a VOP3
// v_fma_f32 (which cannot carry the modifier) -- 128 instead of 64 instructions per slice and step set -- and a DPP source
// must not have been written by the vector ALU within the two preceding wait states, a hazard the compiler pads for its own
// instructions only: the block opens with `s_nop 1` and reads, in each instruction, a register last written sixteen
// instructions earlier (or before the block); no reload can come between them.  Registers 4 q .. 4 q + 3 take gate_a for even
// q and gate_b for odd q (16x16 shape: the two edges of a lane; the 32x32 shapes pass the same gate twice).  Same arithmetic
// as the two-instruction form: one fused multiply-add per value.
#define MDX_FMAC_DPP(R, G) "v_fmac_f32_dpp %" #R ", %" #R ", %" #G " row_shr:%18 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
template <int D>
__device__ __forceinline__ void segmented_step(float (&x)[16], float gate_a, float gate_b)
{
    asm volatile("s_nop 1\n\t"
                 MDX_FMAC_DPP(0, 16) MDX_FMAC_DPP(1, 16) MDX_FMAC_DPP(2, 16) MDX_FMAC_DPP(3, 16)
                 MDX_FMAC_DPP(4, 17) MDX_FMAC_DPP(5, 17) MDX_FMAC_DPP(6, 17) MDX_FMAC_DPP(7, 17)
                 MDX_FMAC_DPP(8, 16) MDX_FMAC_DPP(9, 16) MDX_FMAC_DPP(10, 16) MDX_FMAC_DPP(11, 16)
                 MDX_FMAC_DPP(12, 17) MDX_FMAC_DPP(13, 17) MDX_FMAC_DPP(14, 17) MDX_FMAC_DPP(15, 17)
                 : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
                   "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
                 : "v"(gate_a), "v"(gate_b), "n"(D));
}
#undef MDX_FMAC_DPP

// Raw accumulators of tile t parked in / taken from the sixteen registers that tile occupies in an operand set (MODE 3).
template <int H>
__device__ __forceinline__ void park(Act<H, 0>& a, int t, const f32x16& acc)
{
#pragma unroll
    for (int r = 0; r < 16; ++r) a.v[16 * t + r] = acc[r];
}
template <int H>
__device__ __forceinline__ f32x16 unpark(const Act<H, 0>& a, int t)
{
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = a.v[16 * t + r];
    return acc;
}
template <int H>
__device__ __forceinline__ void park(Act<H, 1>& a, int t, const f32x16& acc)
{
    f32x4 q[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r >> 2][r & 3] = acc[r];
    a.hi[2 * t] = __builtin_bit_cast(half8, q[0]);
    a.hi[2 * t + 1] = __builtin_bit_cast(half8, q[1]);
    a.lo[2 * t] = __builtin_bit_cast(half8, q[2]);
    a.lo[2 * t + 1] = __builtin_bit_cast(half8, q[3]);
}
template <int H>
__device__ __forceinline__ void park(Act<H, 2>& a, int t, const f32x16& acc)
{
    f32x4 q[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) q[r >> 2][r & 3] = acc[r];
    a.hi[0][t] = __builtin_bit_cast(half8, q[0]);
    a.hi[1][t] = __builtin_bit_cast(half8, q[1]);
    a.lo[0][t] = __builtin_bit_cast(half8, q[2]);
    a.lo[1][t] = __builtin_bit_cast(half8, q[3]);
}
template <int H>
__device__ __forceinline__ f32x16 unpark(const Act<H, 2>& a, int t)
{
    const f32x4 q[4] = {__builtin_bit_cast(f32x4, a.hi[0][t]), __builtin_bit_cast(f32x4, a.hi[1][t]),
                        __builtin_bit_cast(f32x4, a.lo[0][t]), __builtin_bit_cast(f32x4, a.lo[1][t])};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = q[r >> 2][r & 3];
    return acc;
}
template <int H>
__device__ __forceinline__ f32x16 unpark(const Act<H, 1>& a, int t)
{
    const f32x4 q[4] = {__builtin_bit_cast(f32x4, a.hi[2 * t]), __builtin_bit_cast(f32x4, a.hi[2 * t + 1]),
                        __builtin_bit_cast(f32x4, a.lo[2 * t]), __builtin_bit_cast(f32x4, a.lo[2 * t + 1])};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = q[r >> 2][r & 3];
    return acc;
}

// MODE 0: the EGNN edge chain (gathered first layer, messages + head out); MODE 2: the same with the messages added up per
// node inside the kernel (piece sums out).  MODE 1: the same pipeline over the rows of a matrix -- out = residual + W_L (SiLU(W_{L-1} ... SiLU(W_1 x + b_1) ...)) + b_L -- used for the per-node MLP of an EGNN layer.
template <int H, int PREC, int MODE, bool ATT = false>
__global__ __launch_bounds__(kWaves* kWave, 1) void egnn_edge_chain_kernel(ChainArgs p)
{
    static_assert(!ATT || MODE == 0 || MODE == 2, "the attention gate belongs to the edge chain");
    extern __shared__ __attribute__((aligned(16))) char lds_raw[];
    // (GUARD: see issue_piece.  The production-size piece-sums instantiations <256, PREC, 2> skip the three extra wait states
    // in front of their requests -- 2.2 % of the launch, A/B in profiles/r03_chain_ablation.md -- which is safe only while they
    // restore no scalar register from a vector register: the BUILD checks that (csrc/Makefile feeds the compiler's resource
    // remarks of this unit to check_chain_resources.py and fails on a scalar-register spill in any of them).  Every other
    // instantiation is guarded.  -DMDX_CHAIN_GUARD_ALL: all guarded.)
#ifdef MDX_CHAIN_GUARD_ALL
    using C = Chain<H, PREC, true>;
#else
    using C = Chain<H, PREC, ATT || !(MODE == 2 && H == 256)>;      // (ATT: always guarded -- those instantiations may spill)
#endif
    // MODE 3 = MODE 1 whose first layer is 2 H -> H: the rows are [h | agg]; chain "layers" 0 and 1 are the two H x H halves of
    // that layer's weight.  Pass A multiplies h by the first half and PARKS the raw accumulators (bias included) in the
    // registers of the other operand set; pass B reloads the operand registers with agg, starts every tile from its parked
    // accumulator, multiplies by the second half and runs the usual epilogue into the slot the parked tile has left.
    constexpr bool ROWS = MODE == 1 || MODE == 3;
    constexpr int NT = C::NT;
    constexpr int STEPS = PREC == 0 ? H / 8 : H / 16;        // k-steps of a tile: 4 f32 MFMAs | 3 f16 MFMAs (32x32x16) | 6 (16x16x32) each
    constexpr bool SPLIT = PREC >= 1;                        // split-f16 arithmetic (either MFMA shape)
    using L = Lay<PREC>;
    constexpr bool W16 = L::W16;
    constexpr int NS = L::NS;                                // edges (rows) per lane: 1 | 2 (16x16 shape)
    constexpr int PFD = STEPS >= 4 ? 2 : 1;                  // weight fragments are read from LDS this many steps ahead
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave), lane = threadIdx.x % kWave;
    // 32x32 shapes: the lane's edge is column lane & 31 and it holds the feature quads of half h = lane >> 5;
    // 16x16 shape: edges 16 cg + (lane & 15), cg = 0, 1, and the feature quads of quarter g = lane >> 4
    const int h = W16 ? lane >> 4 : lane >> 5, col = W16 ? lane & 15 : lane & 31;
    const int layers = p.n_message + p.n_coord;

    lds_c* ring = (lds_c*)lds_raw;
    lds_f* par = (lds_f*)(lds_raw + kRing * C::CHUNK);      // [layers][H] biases | bias_in | w_radial
    lds_f* par_in = par + layers * H;
    lds_f* par_wr = par_in + H;
    lds_f* par_sc = par_wr + H;                             // [kScaleSlots][4]: struct Scale of every packed layer
    // the source nodes of this wavefront's 32 edges (in-kernel message aggregation: where the pieces end)
    __attribute__((address_space(3))) int* seg_src = (__attribute__((address_space(3))) int*)(par_sc + 4 * kScaleSlots) + wave * 32;
    // exact-f32 kernels: the workgroup's maxima of |u| per position (ChainArgs::act_max), behind the source ids
    uint32_t* max_table = (uint32_t*)(lds_raw + kRing * C::CHUNK + sizeof(float) * ((size_t)layers * H + 2 * H + 4 * kScaleSlots) +
                                      sizeof(int) * kWaves * 32);
    // ATT: w_att ln 2 2^-c (c = the exponent the messages are carried with: the gate's logit straight from the carried values)
    // and, behind it, b_att; 16-byte aligned, behind everything else
    lds_f* par_att = (lds_f*)(lds_raw + ((kRing * C::CHUNK + sizeof(float) * ((size_t)layers * H + 2 * H + 4 * kScaleSlots) +
                                          sizeof(int) * kWaves * 32 + sizeof(uint32_t) * kMaxPositions + 15) & ~(size_t)15));
    constexpr int ACT = kActExp<PREC>;
    // the exponent of the activations carried at position q (see kActExp): the caller's, else ACT; exact-f32 kernels: 0
    auto act_exp = [&](int q) -> int {
        if constexpr (!SPLIT) return 0;
        if (MODE == 3 && q == 1) q = 0;
        return p.act_exps ? p.act_exps[q] : ACT;
    };
    auto in_pos = [](int l) -> int { return l; };
    auto out_pos = [](int l) -> int { return MODE == 3 && l == 0 ? 2 : l + 1; };
    // the four factors at the chain's ends (slot kEnds of the scale table, read where they are used):
    //   [0] 2^-c_0 (the first layer's SiLU)   [1] log2(e) 2^c_0 (a caller's row -> carried)
    //   [2] ln 2 2^-c_out (carried -> the caller's: the messages, position n_message; the rows written out, position L)
    //   [3] log2(e) 2^c_{L+1} (the rows read back in front of the projection layers)
    auto end_factor = [&](int i) -> float { return par_sc[4 * kEnds + i]; };

    // (the device-side count arrives through a vector load: made scalar by hand, or every loop bound derived from it -- and
    // with them the whole ring state of the tile loop -- lives in vector registers and is updated by the vector ALU)
    int64_t n_edges = p.n_edges;
    if (p.n_edges_dev) {
        const int64_t dev_count = *p.n_edges_dev;
        const int64_t uniform = ((int64_t)__builtin_amdgcn_readfirstlane((int)(dev_count >> 32)) << 32) |
                                (uint32_t)__builtin_amdgcn_readfirstlane((int)dev_count);
        n_edges = uniform < p.n_edges ? uniform : p.n_edges;
    }
    const int64_t n_tiles = (n_edges + kTileEdges - 1) / kTileEdges;
    // Tile order: workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on XCD b mod 8), each with its own 4-MB
    // L2.  Every XCD takes ONE contiguous eighth of the tiles and its workgroups walk it side by side, so the node rows an
    // XCD gathers at any moment belong to two or three structures (the edges are sorted by source) and stay in its L2
    // beside the 2.4-MB weight image; dealing tile b + k gridDim to workgroup b instead spreads every structure over all
    // eight L2s (6x the unique bytes fetched, PMC of round 2) and pushes the weight image out.
    const int n_xcd = gridDim.x >= 8 ? 8 : 1;
    const int xcd = blockIdx.x % n_xcd, xcd_slot = blockIdx.x / n_xcd;
    const int xcd_wgs = (gridDim.x - xcd + n_xcd - 1) / n_xcd;               // workgroups on this XCD
    const int64_t xcd_tiles = (n_tiles + n_xcd - 1) / n_xcd;
    const int64_t tile_lo = xcd * xcd_tiles, tile_hi = tile_lo + xcd_tiles < n_tiles ? tile_lo + xcd_tiles : n_tiles;
    if (tile_lo + xcd_slot >= tile_hi) return;              // uniform per workgroup

    // a layer's bias in its accumulator's units: log2(e) 2^(a_l + b_l) bias_l (exact: a power of two)
    for (int i = threadIdx.x; i < layers * H; i += kWaves * kWave) {
        const int l = i / H;
        const int a = SPLIT && p.exps ? p.exps[l] : 0;
        par[i] = p.biases[i] * kLog2e * pow2_bits(a + act_exp(in_pos(l)));
    }
    if (threadIdx.x < kEnds) {
        const int l = threadIdx.x;
        const int packed = layers + (!ROWS ? 1 : (MODE == 3 && p.proj_out ? 2 : 0));
        const int a = SPLIT && p.exps && l < packed ? p.exps[l] : 0;
        // operand / output exponents of slot l: a chain layer; the head (its operand: position L); the projection layers
        // behind a node MLP (their operand: position L + 1); unused slots
        const int b = l < layers ? act_exp(in_pos(l)) : (l < packed ? act_exp(!ROWS ? layers : layers + 1) : 0);
        const int b_out = l < layers ? act_exp(out_pos(l)) : 0;
        par_sc[4 * l + 0] = -pow2_bits(-(a + b));
        par_sc[4 * l + 1] = pow2_bits(a + b - b_out);
        par_sc[4 * l + 2] = pow2_bits(-(a + b - b_out));
        par_sc[4 * l + 3] = kLn2 * pow2_bits(-(a + b));
    } else if (threadIdx.x == kEnds) {
        par_sc[4 * kEnds + 0] = pow2_bits(-act_exp(0));
        par_sc[4 * kEnds + 1] = kLog2e * pow2_bits(act_exp(0));
        par_sc[4 * kEnds + 2] = kLn2 * pow2_bits(-act_exp(!ROWS ? p.n_message : layers));
        par_sc[4 * kEnds + 3] = kLog2e * pow2_bits(act_exp(layers + 1));
    }
    if constexpr (!ROWS) {
        for (int i = threadIdx.x; i < H; i += kWaves * kWave) {
            par_in[i] = p.bias_in[i] * kLog2e;
            par_wr[i] = p.w_radial[i] * kLog2e;
        }
    }
    if constexpr (PREC == 0) {
        if (threadIdx.x < kMaxPositions) max_table[threadIdx.x] = 0u;
    }
    if constexpr (ATT) {
        const float to_message = kLn2 * pow2_bits(-act_exp(p.n_message));
        for (int i = threadIdx.x; i < H; i += kWaves * kWave) par_att[i] = p.att_w[i] * to_message;
        if (threadIdx.x == 0) par_att[H] = p.att_b[0];
    }
    __syncthreads();

#ifdef MDX_CHAIN_STAMPS
    void* stamp_buf_ = p.stamp_buf;
#endif
    MDX_STAMP_ALWAYS(20);
    MDX_STAMP_REALTIME(21);
    C ch;
    ch.image = p.image; ch.chunks_total = (layers + (MODE == 3 && p.proj_out ? 2 : 0)) * NT + (!ROWS ? 1 : 0); ch.next_issue = 0; ch.slot_issue = 0; ch.slot_read = 0;
    ch.ring = ring; ch.wave = wave; ch.lane = lane;
    ch.stores_count = 0;
    ch.share = (int)(((unsigned)wave + blockIdx.x / 8u) & 3u) * (C::CHUNK / 4);      // (blockIdx / 8: its slot on its XCD)
    ch.lane16 = (uint32_t)lane * 16u;
#ifdef MDX_CHAIN_VERIFY
    ch.verify_status = p.status;
#endif
#ifdef MDX_CHAIN_STAMPS
    ch.stamp_buf_ = p.stamp_buf;
#endif

    // Weight fragments of a k-step, and the state that flows from one tile to the next: the LDS address of the tile's
    // chunk, its first PFD fragments and its initial accumulator (= the bias), all read beside the previous tile's MFMAs.
    struct Frag {
        f32x4 f;
        half8 hi, lo;
    };
    auto read_frag = [&](const lds_c* w, int s) -> Frag {
        Frag r;
        if constexpr (PREC == 0) {
            r.f = *(const __attribute__((address_space(3))) f32x4*)(w + s * 1024 + lane * 16);
        } else {
            r.hi = *(const __attribute__((address_space(3))) half8*)(w + s * 2048 + lane * 16);
            r.lo = *(const __attribute__((address_space(3))) half8*)(w + s * 2048 + 1024 + lane * 16);
        }
        return r;
    };
    auto read_bias = [&](const lds_f* bias_row) -> f32x16 {
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (bias_row) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *(const __attribute__((address_space(3))) f32x4*)(bias_row + L::fb(q, h));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[4 * q + i] = b4[i];
            }
        }
        return acc;
    };
    // struct Scale of packed layer l, in scalar registers (one LDS read per layer; PREC 0 never uses it)
    auto load_scale = [&](int l) -> Scale {
        Scale sc{-1.0f, 1.0f, 1.0f, kLn2};
        if constexpr (SPLIT) {
            // (four scalar reads ON PURPOSE: with one 16-byte read, hipcc 7.2 hands element 0 to all four v_readfirstlane --
            // only ds_read_b32 of the first float is emitted; seen in the ISA, reproduced in a ten-line kernel)
            const lds_f* q = par_sc + 4 * l;
            const float v[4] = {q[0], q[1], q[2], q[3]};
            sc.neg_c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[0])));
            sc.k = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[1])));
            sc.inv_a = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[2])));
            sc.out = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v[3])));
        }
        return sc;
    };
    const lds_c* w_cur = ch.prime();
    Frag pre[PFD];
#pragma unroll
    for (int s = 0; s < PFD; ++s) pre[s] = read_frag(w_cur, s);
    f32x16 acc_next = read_bias(par);

    bool out_of_range = false;       // split-f16: a non-finite output (see epilogue_elements)
    for (int64_t tile = tile_lo + xcd_slot; tile < tile_hi; tile += xcd_wgs) {
        MDX_STAMP_ALWAYS(9);
        // ---- this lane's edges (or rows): one (32x32 shapes) or two (16x16 shape: column groups 0 and 1) ---------------------
        int64_t e_raw[NS], e[NS];
#pragma unroll
        for (int es = 0; es < NS; ++es) {
            e_raw[es] = tile * kTileEdges + wave * 32 + (W16 ? 16 * es : 0) + col;
            e[es] = e_raw[es] < n_edges ? e_raw[es] : n_edges - 1;
        }
        // (evaluated where it is needed, not carried through the tile as a pair of scalar registers per edge: the 16x16-shape
        // kernels are short of those)
        auto is_live = [&](int es) -> bool { return e_raw[es] < n_edges; };
        // Store instructions a phase that issues `total` of them per wavefront -- total / NS per edge slot, each under
        // `if (is_live(es))` -- has really issued: a slot none of whose lanes is live is skipped as a whole (the ragged last
        // wavefront of a launch with the 16x16 shape can have edges in its first column group only).  The next chunk wait
        // leaves that many of the youngest operations pending (Chain::stores_count); over-counting would let it return
        // before the chunk it waits for has landed.
        auto stores_issued = [&](int total) -> int {
            int count = 0;
#pragma unroll
            for (int es = 0; es < NS; ++es) count += __builtin_amdgcn_ballot_w64(is_live(es)) != 0 ? total / NS : 0;
            return __builtin_amdgcn_readfirstlane(count);
        };
        Act<H, PREC> xa, xb;
        // exact-f32 kernels with ChainArgs::act_max: the largest |u| this lane has written since the last flush_max(); a flush
        // reduces it over the wavefront and raises the position's word (float bits of non-negative values order like integers)
        float amax = 0.0f;
        auto note4 = [&](float y0, float y1, float y2, float y3) {
            if constexpr (PREC == 0)
                amax = __builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(y0), __builtin_fabsf(y1))),
                                       __builtin_fmaxf(__builtin_fabsf(y2), __builtin_fabsf(y3)));
        };
        auto flush_max = [&](int position) {
            if constexpr (PREC == 0) {
                // (into the workgroup's LDS table: one global atomic per position at the end of the kernel, and no global
                // pointer kept in scalar registers through the tile loop)
                float m = amax;
                m = __builtin_fmaxf(m, __shfl_xor(m, 32));
                m = __builtin_fmaxf(m, __shfl_xor(m, 16));
                m = __builtin_fmaxf(m, __shfl_xor(m, 8));
                m = __builtin_fmaxf(m, __shfl_xor(m, 4));
                m = __builtin_fmaxf(m, __shfl_xor(m, 2));
                m = __builtin_fmaxf(m, __shfl_xor(m, 1));
                if (lane == 0) atomicMax(max_table + position, __builtin_bit_cast(uint32_t, m));
                amax = 0.0f;
            }
        };
        // float4 group q8 = 4 t + q of the lane's H / 2 values: registers 4 q .. 4 q + 3 of tile t = features
        // 32 t + fb(q) .. + 3 of edge es(q)  (struct Lay)
        if constexpr (!ROWS) {
            // The top of a tile is a chain of dependent memory latencies with nothing else for the wavefront to do (one per
            // SIMD): edge -> (src, dst) -> rows of the node table.  Everything that depends only on (src, dst) is therefore
            // requested AT ONCE -- both edge slots' pairs as one 16-byte load each; then the first gather batches and every
            // coordinate (the D <= kMaxD loads of a node unrolled, none under a branch) -- and waited for once.  The first form
            // of this block walked the coordinates in a run-time loop with a full wait per component and handled the edge
            // slots one after the other: fourteen latencies in a row per tile (ISA of round 3), ~4 % of the launch.
            typedef int64_t i64x2 __attribute__((ext_vector_type(2)));
            float radial[NS];
            const float *ps[NS], *pd[NS];
            int64_t src[NS], dst[NS];
#pragma unroll
            for (int es = 0; es < NS; ++es) {
                const i64x2 pair = *(const i64x2*)(p.edges + 2 * e[es]);
                src[es] = pair[0];
                dst[es] = pair[1];
                ps[es] = p.node_proj + src[es] * 2 * H;
                pd[es] = p.node_proj + dst[es] * 2 * H + H;
            }
            // first message layer, straight into B-operand registers.
            // The gathers are software-pipelined by hand: batches of QB float4 pairs, requested DEPTH batches before they are
            // used, with scheduling pins between "request" and "compute" (left to itself the compiler keeps six to ten of
            // the 64 loads in flight -- the operand sets fill the register file -- and the phase is a chain of L2 latencies:
            // 25 k cycles per tile against 7 k of arithmetic).
            constexpr int NQ = H / 8, QB = NQ >= 8 ? 4 : NQ, NB = NQ / QB, DEPTH = 4;
            f32x4 ga[NB][QB], gb[NB][QB];
            auto request = [&](int batch) {
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int q8 = batch * QB + k, es = L::es(q8 & 3), off = 32 * (q8 >> 2) + L::fb(q8 & 3, h);
                    ga[batch][k] = *(const f32x4*)(ps[es] + off);
                    gb[batch][k] = *(const f32x4*)(pd[es] + off);
                }
            };
#pragma unroll
            for (int batch = 0; batch < DEPTH && batch < NB; ++batch) request(batch);
            // the squared distance of the edge's end points: sum over k < D of (c_src[k] - c_dst[k])^2, in that order.  D = 6 (the
            // EGNN's torus uplift of three dimensions) is the unrolled form: twelve loads per edge slot at immediate offsets
            // from the two row pointers, all in flight together; any other D walks the components (a latency per component).
            if (p.D == kFastD) {
                float cs[NS][kFastD], cd[NS][kFastD];
#pragma unroll
                for (int es = 0; es < NS; ++es) {
                    const float *rs = p.coord + src[es] * kFastD, *rd = p.coord + dst[es] * kFastD;
#pragma unroll
                    for (int k = 0; k < kFastD; ++k) {
                        cs[es][k] = rs[k];
                        cd[es][k] = rd[k];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int es = 0; es < NS; ++es) {
                    float r2 = 0.0f;
#pragma unroll
                    for (int k = 0; k < kFastD; ++k) {
                        const float dlt = cs[es][k] - cd[es][k];
                        r2 += dlt * dlt;
                    }
                    radial[es] = r2;
                }
            } else {
#pragma unroll
                for (int es = 0; es < NS; ++es) {
                    float r2 = 0.0f;
                    for (int k = 0; k < p.D; ++k) {
                        const float dlt = p.coord[src[es] * p.D + k] - p.coord[dst[es] * p.D + k];
                        r2 += dlt * dlt;
                    }
                    radial[es] = r2;
                }
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int es = 0; es < NS; ++es)
                    if (h == 0) seg_src[(W16 ? 16 * es : 0) + col] = (int)src[es];           // (node indices fit 31 bits: checked on the host)
            }
            __builtin_amdgcn_sched_barrier(0);
            const float kb = end_factor(0);
#pragma unroll
            for (int batch = 0; batch < NB; ++batch) {
                if (batch + DEPTH < NB) request(batch + DEPTH);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < QB; ++k) {
                    const int q8 = batch * QB + k, es = L::es(q8 & 3), off = 32 * (q8 >> 2) + L::fb(q8 & 3, h);
                    const f32x4 a = ga[batch][k], b = gb[batch][k];
                    const f32x4 b0 = *(const __attribute__((address_space(3))) f32x4*)(par_in + off);
                    const f32x4 wr = *(const __attribute__((address_space(3))) f32x4*)(par_wr + off);
                    // z = log2(e) ((a + b) + b0 + radial wr): b0 and wr are staged pre-scaled, two fused multiply-adds per value
                    float y[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) y[i] = silu_first<PREC>(__builtin_fmaf(a[i] + b[i], kLog2e, __builtin_fmaf(radial[es], wr[i], b0[i])), kb);
                    note4(y[0], y[1], y[2], y[3]);
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3), y[0], y[1]);
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3) + 2, y[2], y[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // the rows themselves (the caller's activations, carried as 2^b log2(e) x inside the chain)
            const float kIn = end_factor(1);
#pragma unroll
            for (int q8 = 0; q8 < H / 8; ++q8) {
                const int es = L::es(q8 & 3), off = 32 * (q8 >> 2) + L::fb(q8 & 3, h);
                const f32x4 a = *(const f32x4*)(p.rows_in + e[es] * p.ld_in + off);
                if constexpr (SPLIT) {
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3), a[0] * kIn, a[1] * kIn);
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3) + 2, a[2] * kIn, a[3] * kIn);
                } else {
                    note4(a[0] * kIn, a[1] * kIn, a[2] * kIn, a[3] * kIn);
#pragma unroll
                    for (int i = 0; i < 4; ++i) put<H>(xa, q8 >> 2, 4 * (q8 & 3) + i, a[i] * kIn);
                }
                if ((q8 & 15) == 15) __builtin_amdgcn_sched_barrier(0);
            }
        }
        flush_max(0);
        MDX_STAMP_ALWAYS(10);
        if constexpr (C::SPREAD) ch.scalar_addresses();       // (the burst form sets them right before its requests)
        // ---- the chain ----------------------------------------------------------------------------------------------
        f32x16 pend;                                          // accumulators whose epilogue is outstanding

        // One tile: the MFMAs of (layer, t) from operand registers `in`; beside them (a) the outstanding epilogue, which
        // writes tile `tp` of `epi_dst`, (b) from the middle on, the reads of the NEXT tile's first fragments and bias
        // (next_bias == nullptr: that tile starts from zero -- the head).
        auto run_tile = [&](const Act<H, PREC>& in, bool have, int tp, Act<H, PREC>& epi_dst, const lds_f* next_bias,
                            const Scale& epi_sc, bool linear = false) -> f32x16 {
            MDX_STAMP(4);
            // ATT: the request addresses in scalar registers again at the top of EVERY tile (two v_readfirstlane).  The branch
            // around the gate adds control-flow merges to the layer loop, and a value that reaches a request through such a
            // merge is handed over in vector registers (see scalar_addresses); this way every request of a tile takes its
            // addresses from a statement of the same straight-line stretch -- this one, or the acquire in the tile's middle.
            if constexpr (ATT && C::SPREAD) ch.scalar_addresses();
            f32x16 acc = acc_next;
            constexpr bool STAGED = SPLIT && STEPS == 16;
            EpiloguePipe<H> pipe;
            Frag fr[STEPS + PFD];
#pragma unroll
            for (int s = 0; s < PFD; ++s) fr[s] = pre[s];
            const lds_c* w_next = nullptr;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s == STEPS / 2) w_next = ch.acquire_next();
                if (s + PFD < STEPS) fr[s + PFD] = read_frag(w_cur, s + PFD);
                else fr[s + PFD] = read_frag(w_next, s + PFD - STEPS);
                if (s == STEPS - 1) acc_next = read_bias(next_bias);
                if constexpr (STAGED) {
                    if (have) pipe.template step<PREC>(s, tp, pend, epi_dst, epi_sc, ROWS && linear);
                } else if constexpr (W16) {
                    // (16x16 shape: ALL sixteen values of the pending tile are elements of ONE k-step of the next layer, the one
                    // this tile reads in its last two steps when it is that layer's first tile: complete before those steps)
                    constexpr int D = STEPS > 2 ? STEPS - 2 : 1;
                    if (have && s < D) epilogue_elements<H, PREC>(tp, 16 * s / D, 16 * (s + 1) / D, pend, epi_dst, epi_sc, ROWS && linear, amax);
                } else {
                    if (have) epilogue_elements<H, PREC>(tp, 16 * s / STEPS, 16 * (s + 1) / STEPS, pend, epi_dst, epi_sc, ROWS && linear, amax);
                }
                // one weight-stream request per STEPS / LPW k-steps, behind the step's first MFMA; g = steps since the acquire
                constexpr int PERIOD = STEPS / C::LPW;
                const int g = (s + STEPS - STEPS / 2) % STEPS;
                if constexpr (PREC == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fr[s].f[i], in.v[4 * s + i], acc, 0, 0, 0);
                        if (C::SPREAD && i == 0 && g % PERIOD == 0) ch.issue_piece(g / PERIOD);
                    }
                } else if constexpr (W16) {
                    // step s = 2 ks + rho: the fragment of row tile rho (16 rows x 32 k) against k-step ks of both column groups;
                    // the same A operand four times in a row, the two accumulators alternating
                    const int ks = s >> 1, rho = s & 1;
                    f32x4 a0 = {acc[8 * rho], acc[8 * rho + 1], acc[8 * rho + 2], acc[8 * rho + 3]};
                    f32x4 a1 = {acc[8 * rho + 4], acc[8 * rho + 5], acc[8 * rho + 6], acc[8 * rho + 7]};
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].hi, in.hi[0][ks], a0, 0, 0, 0);
                    if (C::SPREAD && g % PERIOD == 0) ch.issue_piece(g / PERIOD);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].hi, in.hi[1][ks], a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].hi, in.lo[0][ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].hi, in.lo[1][ks], a1, 0, 0, 0);
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].lo, in.hi[0][ks], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[s].lo, in.hi[1][ks], a1, 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[8 * rho + i] = a0[i];
                        acc[8 * rho + 4 + i] = a1[i];
                    }
                } else {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[s].hi, in.hi[s], acc, 0, 0, 0);
                    if (C::SPREAD && g % PERIOD == 0) ch.issue_piece(g / PERIOD);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[s].hi, in.lo[s], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[s].lo, in.hi[s], acc, 0, 0, 0);
                }
                if constexpr (STAGED) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < PFD; ++s) pre[s] = fr[STEPS + s];
            w_cur = w_next;
            // keep every tile's share of vector work beside ITS MFMAs: left free, the scheduler sinks the epilogues of the
            // first tiles of a layer into its last ones (their results are not needed before the next layer), which then
            // carry twice the vector work and are bound by instruction issue
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("; MDX_TILE_END" ::);          // (a comment in the assembly: what tools/tile_stats.py cuts the listing at)
            return acc;
        };
        // One layer: tiles t = 0 .. NT-1.  The epilogue beside tile t is that of the tile before it: tile t-1 of this layer
        // (-> out), or, at t = 0, the last tile of the previous layer (-> in: its features are the last k-steps of this
        // layer, produced before the MFMAs that read them).  FIRST: nothing is outstanding at t = 0.
        Scale sc_prev = load_scale(0);                        // of the layer whose last tile's epilogue is outstanding
        auto layer = [&](auto first_tag, Act<H, PREC>& in, Act<H, PREC>& out, int l) {
            constexpr bool FIRST = decltype(first_tag)::value;
            const lds_f* bias = par + l * H;
            const Scale sc = load_scale(l);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                // the tile after this one: the next tile of this layer, the first of the next layer, or -- after the last
                // layer -- the head (no bias; MODE 0) / layer 0 of the next rows (MODE 1)
                const lds_f* next_bias = t + 1 < NT ? bias + 32 * (t + 1) : (l + 1 < layers ? bias + H : (!ROWS || (MODE == 3 && p.proj_out) ? nullptr : par));
                // the epilogue beside tile 0 belongs to the previous layer (never the linear one); the others to this layer
                if (t == 0) {
                    pend = run_tile(in, !FIRST, NT - 1, in, next_bias, sc_prev);
                    if (!FIRST) flush_max(l);               // the operand of layer l is complete
                } else {
                    pend = run_tile(in, true, t - 1, out, next_bias, sc, l == layers - 1);
                }
            }
            sc_prev = sc;
        };
        // ATT.  The messages must be gated BEFORE the first coordinate layer reads them, so the pipelining across that one
        // layer boundary is given up: the last message tile's epilogue is run at once (not beside the next layer's first
        // tile), then per edge  a = m . w_att + b_att  (this lane's features, then the other lanes of the edge: 32x32 shapes
        // lane ^ 32; 16x16 shape lanes ^ 16 and ^ 32),  g = 1 / (1 + e^-a),  and every carried value of the edge is multiplied
        // by g and written back into the operand registers (split-f16: split again).  The first coordinate layer then starts
        // like the chain's first layer (nothing outstanding).  H / 2 fused multiply-adds and as many products per lane: noise
        // beside the layer's 2 H^2 MACs per edge.
        auto gate_messages = [&](Act<H, PREC>& m, int l) {
            if constexpr (ATT) {
                epilogue_elements<H, PREC>(NT - 1, 0, 16, pend, m, sc_prev, false, amax);
                flush_max(l);
                float part[NS];
#pragma unroll
                for (int es = 0; es < NS; ++es) part[es] = 0.0f;
#pragma unroll
                for (int q8 = 0; q8 < H / 8; ++q8) {
                    const int t = q8 >> 2, q = q8 & 3;
                    const f32x4 y = get4<H>(m, t, q);
                    const f32x4 w4 = *(const __attribute__((address_space(3))) f32x4*)(par_att + 32 * t + L::fb(q, h));
#pragma unroll
                    for (int i = 0; i < 4; ++i) part[L::es(q)] = __builtin_fmaf(y[i], w4[i], part[L::es(q)]);
                }
                float g[NS];
#pragma unroll
                for (int es = 0; es < NS; ++es) {
                    float a = part[es];
                    if constexpr (W16) a += __shfl_xor(a, 16);
                    a += __shfl_xor(a, 32);
                    a += par_att[H];
                    g[es] = 1.0f / (1.0f + expf(-a));
                }
#pragma unroll
                for (int q8 = 0; q8 < H / 8; ++q8) {
                    const int t = q8 >> 2, q = q8 & 3;
                    const f32x4 y = get4<H>(m, t, q);
                    const float gg = g[L::es(q)];
                    put_pair<H>(m, t, 4 * q, y[0] * gg, y[1] * gg);
                    put_pair<H>(m, t, 4 * q + 2, y[2] * gg, y[3] * gg);
                    if ((q8 & 15) == 15) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // messages = the operand registers of the first coordinate layer, complete once its first tile has run
        auto store_messages = [&](const Act<H, PREC>& m) {
            // every wavefront issues exactly H / 8 store instructions (lanes beyond the edge count masked off, the address
            // clamped; an edge slot without a live lane issues none): the next chunk wait counts on them (Chain::stores_count)
            const float kOut = end_factor(2);
#pragma unroll
            for (int q8 = 0; q8 < H / 8; ++q8) {
                const int t = q8 >> 2, q = q8 & 3, es = L::es(q);
                f32x4 y = get4<H>(m, t, q);
#pragma unroll
                for (int i = 0; i < 4; ++i) y[i] *= kOut;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) out_of_range = out_of_range || (is_live(es) && !(__builtin_fabsf(y[i]) <= 3.0e38f));
                }
                if (is_live(es)) *(f32x4*)(p.messages + e[es] * H + 32 * t + L::fb(q, h)) = y;
            }
            ch.stores_count = stores_issued(H / 8);
        };
        // Message aggregation inside the kernel (piece_sums): the edges are sorted by source node, so a node's edges are a
        // run of consecutive edges = consecutive lanes of the accumulator layout (32x32 shapes: lane = 32 h + edge; 16x16 shape:
        // lane = 16 g + edge within column group cg).  A PIECE = a node's edges inside one 16-edge group = inside one DPP row;
        // its sum is a segmented scan over the row, four shift-and-add steps (`row_shr` 1, 2, 4, 8, each gated by "the edge
        // that far down is in my piece") on the registers where the messages already are -- no staging, no transposition.  The
        // lane of a piece's LAST edge then holds the piece sum of its feature quads and writes them to the piece's row of the
        // compact buffer; mdx_segment_combine adds a node's pieces in edge order: fixed order, no atomics.  The messages
        // themselves never reach memory.  Returns a lower bound of the store instructions issued (for the next chunk wait).
        auto aggregate_pieces = [&](const Act<H, PREC>& m) -> int {
            const int64_t wave_base = tile * kTileEdges + wave * 32;
            const float kOut = end_factor(2);
            const int n_live = n_edges - wave_base >= 32 ? 32 : (n_edges > wave_base ? (int)(n_edges - wave_base) : 0);
            const int c16 = col & 15;
            float gate[NS][4];                              // 1.0: the edge 1 / 2 / 4 / 8 below is in this lane's piece
            bool ends[NS];
            float* row[NS];
#pragma unroll
            for (int es = 0; es < NS; ++es) {
                const int pos = (W16 ? 16 * es : 0) + col;          // position among the wavefront's 32 edges
                const int my_src = seg_src[pos];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = 1 << k;
                    gate[es][k] = (c16 >= d && seg_src[c16 >= d ? pos - d : pos] == my_src) ? 1.0f : 0.0f;
                }
                ends[es] = pos < n_live && (c16 == 15 || pos + 1 >= n_live || seg_src[pos + 1 < 32 ? pos + 1 : pos] != my_src);
                // Compact piece rows (no [E, H] buffer): a piece that ends on a 16-edge boundary goes to row e / 16; any other
                // piece end is the LAST edge of its node (the next edge has another source, and it is inside this wavefront's
                // 32 edges because its position is not 15 mod 16) and goes to the node's own row behind the ceil(capacity / 16)
                // boundary rows.
                const int64_t e_mine = wave_base + pos;
                const int64_t piece_row = c16 == 15 ? (e_mine >> 4) : ((p.n_edges + 15) >> 4) + (int64_t)my_src;
                row[es] = p.messages + piece_row * H;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float x[16];                                // the sixteen values of tile t, as they lie in the accumulator layout
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 y = get4<H>(m, t, q);
#pragma unroll
                    for (int i = 0; i < 4; ++i) x[4 * q + i] = y[i] * kOut;
                }
                // (registers 4 q .. 4 q + 3 belong to edge es(q): q & 1 on the 16x16 shape, the lane's one edge otherwise)
                segmented_step<1>(x, gate[0][0], gate[NS - 1][0]);
                segmented_step<2>(x, gate[0][1], gate[NS - 1][1]);
                segmented_step<4>(x, gate[0][2], gate[NS - 1][2]);
                segmented_step<8>(x, gate[0][3], gate[NS - 1][3]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    // (split-f16: no range check here -- a message beyond the f16 range is an infinity or a NaN in
                    // the operand registers the coordinate layers read next, and reaches this edge's head output,
                    // which is checked)
                    if (ends[L::es(q)]) {
                        const f32x4 y = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
                        *(f32x4*)(row[L::es(q)] + 32 * t + L::fb(q, h)) = y;
                    }
                }
                __builtin_amdgcn_sched_barrier(0);         // one slice at a time: the register file is full here
            }
            // store instructions this wavefront issued: per edge slot H / (8 NS), each with the piece-end lanes active (none: none)
            int count = 0;
#pragma unroll
            for (int es = 0; es < NS; ++es) count += __builtin_amdgcn_ballot_w64(ends[es]) != 0 ? H / (8 * NS) : 0;
            return count;
        };
        // The aggregation phase needs registers of its own while both operand sets are live: the next tile's prefetched
        // fragments and initial accumulator (32 registers) are simply read again from LDS afterwards instead of being kept.
        auto reload_prefetch = [&](const lds_f* bias_row) {
#pragma unroll
            for (int s2 = 0; s2 < PFD; ++s2) pre[s2] = read_frag(w_cur, s2);
            acc_next = read_bias(bias_row);
        };
        // the head: one more tile whose image row 0 is w_out (no bias, no activation): s_e = row 0 of the accumulator --
        // register 0 of the lanes with h == 0 (16x16 shape: registers 0 and 4, the two column groups)
        auto head_tile = [&](Act<H, PREC>& in) {
            const f32x16 acc = run_tile(in, true, NT - 1, in, par, sc_prev);      // after the head: layer 0, tile 0 of the next edges
            flush_max(layers);
            const float to_out = load_scale(layers).out;                          // (the head row is packed as it is: ln 2 comes here)
#pragma unroll
            for (int es = 0; es < NS; ++es) {
                const float s_e = acc[4 * es] * to_out;
                if constexpr (SPLIT) out_of_range = out_of_range || (is_live(es) && !(__builtin_fabsf(s_e) <= 3.0e38f));
                if (is_live(es) && h == 0) p.edge_scalar[e[es]] = s_e;
            }
        };
        // MODE 1: the last tile of the last (linear) layer has no tile after it to run beside; then out = residual + y
        auto finish_rows = [&](Act<H, PREC>& y, Act<H, PREC>& u) {
            epilogue_elements<H, PREC>(NT - 1, 0, 16, pend, y, sc_prev, true, amax);
            flush_max(layers);
            const bool project = MODE == 3 && p.proj_out != nullptr;       // out is also the operand of two more linear layers
            const float kOut = end_factor(2);
#pragma unroll
            for (int q8 = 0; q8 < H / 8; ++q8) {
                const int t = q8 >> 2, q = q8 & 3, es = L::es(q), off = 32 * t + L::fb(q, h);
                f32x4 v = get4<H>(y, t, q);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= kOut;
                if constexpr (SPLIT) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) out_of_range = out_of_range || (is_live(es) && !(__builtin_fabsf(v[i]) <= 3.0e38f));
                }
                if (is_live(es)) {
                    if (p.residual) {
                        const f32x4 r4 = *(const f32x4*)(p.residual + e[es] * p.ld_in + off);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] = r4[i] + v[i];
                    }
                    *(f32x4*)(p.rows_out + e[es] * H + off) = v;
                }
                if ((q8 & 7) == 7) __builtin_amdgcn_sched_barrier(0);      // (bounds the hoisting of the residual loads)
            }
            ch.stores_count = stores_issued(H / 8);
            if constexpr (MODE == 3) {
                if (project) {
                    // [out W_src^T | out W_dst^T]: 2 NT more tiles, each stored as it is (sixteen tiles per 128 rows: not worth
                    // a pipelined epilogue).  Their operand: the rows just written, read back into the free register set
                    // (filling it while y and the residual are live spills 480 B per lane); this lane reads what it wrote.
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const float kIn = end_factor(3);
#pragma unroll
                    for (int q8 = 0; q8 < H / 8; ++q8) {
                        const int es = L::es(q8 & 3), off = 32 * (q8 >> 2) + L::fb(q8 & 3, h);
                        const f32x4 a = *(const f32x4*)(p.rows_out + e[es] * H + off);
                        note4(a[0] * kIn, a[1] * kIn, a[2] * kIn, a[3] * kIn);
                        put_pair<H>(u, q8 >> 2, 4 * (q8 & 3), a[0] * kIn, a[1] * kIn);
                        put_pair<H>(u, q8 >> 2, 4 * (q8 & 3) + 2, a[2] * kIn, a[3] * kIn);
                        if ((q8 & 15) == 15) __builtin_amdgcn_sched_barrier(0);
                    }
                    flush_max(layers + 1);
#pragma unroll 1
                    for (int t2 = 0; t2 < 2 * NT; ++t2) {
                        if constexpr (C::SPREAD) ch.scalar_addresses();    // (loop-carried: see scalar_addresses)
                        const f32x16 acc = run_tile(u, false, 0, u, t2 + 1 < 2 * NT ? nullptr : par, sc_prev);
                        const float to_out = load_scale(layers + t2 / NT).out;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int es = L::es(q);
                            const f32x4 v = {acc[4 * q] * to_out, acc[4 * q + 1] * to_out, acc[4 * q + 2] * to_out, acc[4 * q + 3] * to_out};
                            if constexpr (SPLIT) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) out_of_range = out_of_range || (is_live(es) && !(__builtin_fabsf(v[i]) <= 3.0e38f));
                            }
                            if (is_live(es)) *(f32x4*)(p.proj_out + e[es] * 2 * H + 32 * t2 + L::fb(q, h)) = v;
                        }
                        ch.stores_count = stores_issued(4);
                    }
                }
            }
        };
        int l_first = 1;
        if constexpr (MODE == 3) {
            // pass A: layer 0 = the first half of the wide weight on h; no epilogue, the accumulators are parked in xb
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x16 acc = run_tile(xa, false, 0, xa, t + 1 < NT ? par + 32 * (t + 1) : nullptr, sc_prev);
                park(xb, t, acc);
            }
            // the operand registers again, with agg (columns H .. 2H-1 of the row)
            const float kIn = end_factor(1);
#pragma unroll
            for (int q8 = 0; q8 < H / 8; ++q8) {
                const int es = L::es(q8 & 3), off = 32 * (q8 >> 2) + L::fb(q8 & 3, h);
                const f32x4 a = *(const f32x4*)(p.rows_in2 + e[es] * p.ld_in2 + off);
                if constexpr (SPLIT) {
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3), a[0] * kIn, a[1] * kIn);
                    put_pair<H>(xa, q8 >> 2, 4 * (q8 & 3) + 2, a[2] * kIn, a[3] * kIn);
                } else {
                    note4(a[0] * kIn, a[1] * kIn, a[2] * kIn, a[3] * kIn);
#pragma unroll
                    for (int i = 0; i < 4; ++i) put<H>(xa, q8 >> 2, 4 * (q8 & 3) + i, a[i] * kIn);
                }
                if ((q8 & 15) == 15) __builtin_amdgcn_sched_barrier(0);
            }
            flush_max(0);
            // pass B: layer 1 = the second half on agg, every tile started from its parked accumulator
            acc_next = unpark(xb, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const lds_f* after = t + 1 < NT ? nullptr : par + 2 * H;          // layer 2, tile 0 (there always is one)
                pend = run_tile(xa, t > 0, t - 1, xb, after, sc_prev);      // (layers 0 and 1 share one exponent: sc_prev = layer 0's)
                if (t + 1 < NT) acc_next = unpark(xb, t + 1);
            }
            l_first = 2;
        } else {
            layer(std::true_type{}, xa, xb, 0);
        }
        bool done = false;
        if constexpr (MODE == 1) {                      // (the edge chain always has a message and a coordinate layer)
            if (layers == 1) {
                finish_rows(xb, xa);
                done = true;
            }
        }
        if (!done) {
            // layer l from `in` to `out`; ATT: the first coordinate layer starts behind the gate, with nothing outstanding
            auto step_layer = [&](Act<H, PREC>& in, Act<H, PREC>& out, int l) {
                if constexpr (ATT) {
                    if (l == p.n_message) {
                        gate_messages(in, l);
                        // (like the aggregation phase: the next tile's prefetched fragments and initial accumulator are read
                        // again from LDS behind the gate instead of being kept live across it -- 48 registers the gate needs)
                        reload_prefetch(par + l * H);
                        layer(std::true_type{}, in, out, l);
                    } else {
                        layer(std::false_type{}, in, out, l);
                    }
                } else {
                    layer(std::false_type{}, in, out, l);
                }
            };
            for (int l = l_first;;) {
                step_layer(xb, xa, l);
                if (!ROWS && l == p.n_message) {
                    if constexpr (MODE == 2) {
                        MDX_STAMP_ALWAYS(30);
                        ch.stores_count = aggregate_pieces(xb);
                        reload_prefetch(l + 1 < layers ? par + (l + 1) * H : nullptr);
                        MDX_STAMP_ALWAYS(31);
                    } else {
                        store_messages(xb);
                    }
                }
                if (++l >= layers) {
                    if constexpr (!ROWS) head_tile(xa);
                    else finish_rows(xa, xb);
                    break;
                }
                step_layer(xa, xb, l);
                if (!ROWS && l == p.n_message) {
                    if constexpr (MODE == 2) {
                        MDX_STAMP_ALWAYS(30);
                        ch.stores_count = aggregate_pieces(xa);
                        reload_prefetch(l + 1 < layers ? par + (l + 1) * H : nullptr);
                        MDX_STAMP_ALWAYS(31);
                    } else {
                        store_messages(xa);
                    }
                }
                if (++l >= layers) {
                    if constexpr (!ROWS) head_tile(xb);
                    else finish_rows(xb, xa);
                    break;
                }
            }
        }
    }
    // requests still in flight target this workgroup's LDS: let them land before the workgroup ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MDX_STAMP_ALWAYS(22);
    MDX_STAMP_REALTIME(23);
    if constexpr (SPLIT) {
        if (p.status && out_of_range) atomicOr(p.status, MDX_STATUS_EGNN_F16_RANGE);
    } else {
        __syncthreads();
        if (p.act_max && (int)threadIdx.x < layers + 2 && max_table[threadIdx.x]) atomicMax(p.act_max + threadIdx.x, max_table[threadIdx.x]);
    }
}

// ---- weight image ------------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w[MDX_EGNN_CHAIN_MAX_LAYERS];      // [H][H] nn.Linear weights (out, in), device
    const float* w_out;                             // [H] the coordinate head: row 0 of one more 32-row chunk (rest zero)
    int layers, H, precision;
    uint32_t tied;                                  // bit l: layer l shares its exponent with layer l - 1
    int32_t* exps;                                  // [layers + 1] (split-f16): the image holds 2^exps[l] W_l
    void* image;
};

// Split-f16 exponents, pass 1: exps[l] (as unsigned) = the bits of max |W_l| (non-negative floats order like their bits).
__global__ __launch_bounds__(256) void egnn_chain_maxabs_kernel(PackArgs p)
{
    const int l = blockIdx.y;                       // == p.layers: the head row
    const bool head = l == p.layers;
    const float* w = head ? p.w_out : p.w[l];
    const int64_t n = head ? p.H : (int64_t)p.H * p.H;
    uint32_t m = 0;
    if (w)
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
            const uint32_t b = __builtin_bit_cast(uint32_t, w[i]) & 0x7fffffffu;
            if (b < 0x7f800000u && b > m) m = b;    // (infinities and NaNs are not magnitudes to scale for)
        }
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)m, d);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax((uint32_t*)p.exps + l, m);
}
// pass 2 (one thread): bits of the maximum -> a_l with 2^a_l max|W_l| in [2^13, 2^14) (0 for an all-zero layer); tied layers
// (the two halves of a wide first layer, whose accumulators continue one another) take the smaller exponent of their run.
__global__ void egnn_chain_exponents_kernel(PackArgs p)
{
    if (blockIdx.x || threadIdx.x) return;
    for (int l = 0; l <= p.layers; ++l) {
        const uint32_t b = ((const uint32_t*)p.exps)[l];
        int a = 0;
        if (b) {
            const int e = (int)(b >> 23) - 127;     // floor(log2 m) for a normal m; subnormal maxima: treated as 2^-126
            a = 13 - (e < -126 ? -126 : e);
            a = a > 40 ? 40 : (a < -40 ? -40 : a);
        }
        p.exps[l] = a;
    }
    for (int l = 1; l < p.layers; ++l)
        if (p.tied >> l & 1u) {
            int lo = l - 1;
            while (lo > 0 && (p.tied >> lo & 1u)) --lo;
            int a = p.exps[lo];
            for (int k = lo + 1; k <= l; ++k) a = p.exps[k] < a ? p.exps[k] : a;
            for (int k = lo; k <= l; ++k) p.exps[k] = a;
        }
}

// Activation exponents from the maxima the exact-f32 kernels collected (one thread): position q with a recorded maximum m
// (float bits) gets min(its current exponent, c) with 2^c m in [2^12, 2^13) -- eight times below the f16 range -- c clamped to
// [kMinActExp, kActExp]: an exponent only ever goes DOWN (a layer seen hot once keeps its headroom), never above the default
// (whose precision floor the accuracy tests hold), and a position nothing was recorded for keeps what it has.  An infinite or
// NaN maximum counts as the largest float.  The maxima are cleared for the next collection.
constexpr int kMinActExp = -14;
__global__ void egnn_chain_act_exponents_kernel(uint32_t* maxima, int count, int32_t* exps)
{
    if (blockIdx.x || threadIdx.x) return;
    for (int q = 0; q < count; ++q) {
        const uint32_t b = maxima[q] & 0x7fffffffu;
        maxima[q] = 0;
        if (!b) continue;
        const int e = b >= 0x7f800000u ? 127 : (int)(b >> 23) - 127;          // floor(log2 m); subnormals: -127
        int c = 12 - e;
        c = c > kActExp<1> ? kActExp<1> : (c < kMinActExp ? kMinActExp : c);
        if (c < exps[q]) exps[q] = c;
    }
}

__global__ __launch_bounds__(256) void egnn_chain_pack_kernel(PackArgs p)
{
    const int H = p.H, NT = H / 32;
    const int64_t per_layer = (int64_t)H * H;
    const int64_t total = per_layer * p.layers + 32 * H;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(idx / per_layer);               // == p.layers: the head chunk
        int64_t r = idx - l * per_layer;
        const int t = (int)(r / (32 * H));                  // chunk = accumulator tile
        r -= (int64_t)t * 32 * H;
        (void)NT;
        const bool head = l == p.layers;
        auto weight = [&](int n, int k) -> float {
            if (head) return (n == 0 && p.w_out) ? p.w_out[k] : 0.0f;       // (the kernel multiplies the head's sum by ln 2)
            return p.w[l][(int64_t)n * H + k];
        };
        if (p.precision == 0) {
            // chunk: [q = H/8][lane 64][4 floats]; lane (n = 32 t + (lane & 31), h = lane >> 5) holds W[n][8 q + 4 h + i]
            const int q = (int)(r / 256), lane = (int)(r % 256) / 4, i = (int)(r % 4);
            const int n = 32 * t + (lane & 31), k = 8 * q + 4 * (lane >> 5) + i;
            ((float*)p.image)[idx] = weight(n, k);
        } else {
            // chunk: [s = H/16][hi: lane 64 x 8 halfs | lo: lane 64 x 8 halfs];
            //   precision 1 (32x32x16): lane = row 32 t + (lane & 31), element j: k = 16 s + 8 (j >> 2) + 4 (lane >> 5) + (j & 3)
            //   precision 2 (16x16x32): fragment s = 2 ks + rho: row 32 t + 16 rho + (lane & 15),
            //                           element j: k = 32 ks + 16 (j >> 2) + 4 (lane >> 4) + (j & 3)
            const int s = (int)(r / 512), lane = (int)(r % 512) / 8, j = (int)(r % 8);
            const int n = p.precision == 1 ? 32 * t + (lane & 31) : 32 * t + 16 * (s & 1) + (lane & 15);
            const int k = p.precision == 1 ? 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3)
                                           : 32 * (s >> 1) + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
            const float v = weight(n, k) * pow2_bits(p.exps[l]);      // exact (a power of two; |v| < 2^14)
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            _Float16* chunk = (_Float16*)((char*)p.image + ((int64_t)l * (H / 32) + t) * ((int64_t)H * 32 * 4));
            chunk[s * 1024 + lane * 8 + j] = hi;
            chunk[s * 1024 + 512 + lane * 8 + j] = lo;
        }
    }
}

// The two per-edge options of E_GCL's coordinate update (models/egnn.py:128-131, 234-264), applied where the head's scalar meets
// the coordinate difference:
//   MDX_EGNN_COORD_TANH       the coordinate MLP ends in nn.Tanh: s_e <- tanh(s_e)
//   MDX_EGNN_COORD_NORMALIZE  coord_diff <- f(r^2) coord_diff, f(r^2) = tanh(r^2) / sqrt(r^2 + epsilon^2), epsilon = 1e-8, r^2 the
//                             squared length of coord_diff summed in component order (normalize_radial_norm)
// trans = (f coord_diff) s, in the reference's order of operations.
__device__ __forceinline__ float coord_head_value(float s, int flags) { return (flags & MDX_EGNN_COORD_TANH) ? tanhf(s) : s; }
__device__ __forceinline__ float normalize_factor(float r2) { return tanhf(r2) / sqrtf(r2 + 1.0e-16f); }

// coord_out[i,:] = coord[i,:] + scale_i sum_{e in segment i} (coord[i,:] - coord[dst_e,:]) * s_e
__global__ __launch_bounds__(256) void egnn_coord_aggregate_kernel(const float* __restrict__ s, const float* __restrict__ coord,
                                                                   const int64_t* __restrict__ edges,
                                                                   const int64_t* __restrict__ offsets,
                                                                   const int64_t* __restrict__ degree, int64_t n_nodes, int D,
                                                                   int mean, int flags, float* __restrict__ out)
{
    const int64_t total = n_nodes * D;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t node = idx / D;
        const int k = (int)(idx - node * D);
        const int64_t e0 = offsets[node], deg = degree[node];
        const float ci = coord[idx];
        float acc = 0.0f;
        for (int64_t e = e0; e < e0 + deg; ++e) {
            const int64_t dst = edges[2 * e + 1];
            float diff = ci - coord[dst * D + k];
            if (flags & MDX_EGNN_COORD_NORMALIZE) {
                float r2 = 0.0f;
                for (int j = 0; j < D; ++j) {
                    const float dj = coord[node * D + j] - coord[dst * D + j];
                    r2 += dj * dj;
                }
                diff = normalize_factor(r2) * diff;
            }
            acc += diff * coord_head_value(s[e], flags);
        }
        if (mean && deg > 0) acc = acc / (float)deg;          // (a true division, as unsorted_segment_mean's: egnn_utils.py:66-68)
        out[idx] = ci + acc;
    }
}

// out[i,:] = (1/degree_i if mean) x sum of node i's pieces, in increasing edge order: the boundary rows e / 16 of its edges
// e = 15 mod 16 and, when its last edge is not one of those, its own row boundary_rows + i (the compact layout the edge
// chain writes, aggregate_pieces).  One wavefront per node, 16 bytes per lane.
// `left` (nullable): out rows are [left[i,:] | sum_i] of width 2 H -- the input of the node MLP's first layer (h | agg), written
// in the one pass over the nodes instead of a concatenation afterwards.
__global__ __launch_bounds__(256) void segment_combine_kernel(const float* __restrict__ pieces, const int64_t* __restrict__ offsets,
                                                              const int64_t* __restrict__ degree, int64_t n_nodes, int H,
                                                              int mean, float* __restrict__ out, const float* __restrict__ left,
                                                              int64_t boundary_rows)
{
    const int lane = threadIdx.x % kWave;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
    const int quads = H >> 2;
    for (int64_t node = wave; node < n_nodes; node += n_waves) {
        const int64_t e0 = offsets[node], deg = degree[node], e1 = e0 + deg;
        const float count = (mean && deg > 0) ? (float)deg : 1.0f;      // the mean DIVIDES by the count (egnn_utils.py:66-68)
        // the node's last piece: a boundary row if its last edge is 15 mod 16, else the node's own row
        const int64_t last_row = ((e1 - 1) & 15) == 15 ? ((e1 - 1) >> 4) : boundary_rows + node;
        for (int q = lane; q < quads; q += kWave) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int64_t e = e0 | 15; e < e1 - 1; e += 16) acc += reinterpret_cast<const f32x4*>(pieces + (e >> 4) * H)[q];
            if (deg > 0) acc += reinterpret_cast<const f32x4*>(pieces + last_row * H)[q];
            if (mean) acc = acc / count;
            if (left) {
                reinterpret_cast<f32x4*>(out + node * 2 * H)[q] = reinterpret_cast<const f32x4*>(left + node * H)[q];
                reinterpret_cast<f32x4*>(out + node * 2 * H + H)[q] = acc;
            } else {
                reinterpret_cast<f32x4*>(out + node * H)[q] = acc;
            }
        }
    }
}

// segment_combine_kernel and egnn_coord_aggregate_kernel as ONE pass over the nodes (one wavefront per node): the message part
// exactly as segment_combine_kernel; the coordinate part with the node's edges dealt to the lanes (edge offset + lane, + 64, ...),
// D partial sums per lane and a butterfly over the wavefront -- a fixed order, no atomics.  D <= 8.
__global__ __launch_bounds__(256) void egnn_node_gather_kernel(const float* __restrict__ pieces, const int64_t* __restrict__ offsets,
                                                               const int64_t* __restrict__ degree, int64_t n_nodes, int H,
                                                               int mean_messages, float* __restrict__ out,
                                                               const float* __restrict__ left, int64_t boundary_rows,
                                                               const float* __restrict__ s, const float* __restrict__ coord,
                                                               const int64_t* __restrict__ edges, int D, int mean_coords,
                                                               int flags, float* __restrict__ coord_out)
{
    const int lane = threadIdx.x % kWave;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) / kWave;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) / kWave;
    const int quads = H >> 2;
    for (int64_t node = wave; node < n_nodes; node += n_waves) {
        const int64_t e0 = offsets[node], deg = degree[node], e1 = e0 + deg;
        // coordinates first (their loads are the long-latency ones)
        float part[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ci[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ci[k] = k < D ? coord[node * D + k] : 0.0f;
        for (int64_t e = e0 + lane; e < e1; e += kWave) {
            const int64_t dst = edges[2 * e + 1];
            const float se = coord_head_value(s[e], flags);
            if (flags & MDX_EGNN_COORD_NORMALIZE) {          // (uniform per launch)
                float diff[8], r2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    diff[k] = k < D ? ci[k] - coord[dst * D + k] : 0.0f;
                    if (k < D) r2 += diff[k] * diff[k];
                }
                const float f = normalize_factor(r2);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < D) part[k] += (f * diff[k]) * se;
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (k < D) part[k] += (ci[k] - coord[dst * D + k]) * se;
            }
        }
        // messages
        const float count = (mean_messages && deg > 0) ? (float)deg : 1.0f;      // (a true division: egnn_utils.py:66-68)
        const int64_t last_row = ((e1 - 1) & 15) == 15 ? ((e1 - 1) >> 4) : boundary_rows + node;
        for (int q = lane; q < quads; q += kWave) {
            f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int64_t e = e0 | 15; e < e1 - 1; e += 16) acc += reinterpret_cast<const f32x4*>(pieces + (e >> 4) * H)[q];
            if (deg > 0) acc += reinterpret_cast<const f32x4*>(pieces + last_row * H)[q];
            if (mean_messages) acc = acc / count;
            if (left) {
                reinterpret_cast<f32x4*>(out + node * 2 * H)[q] = reinterpret_cast<const f32x4*>(left + node * H)[q];
                reinterpret_cast<f32x4*>(out + node * 2 * H + H)[q] = acc;
            } else {
                reinterpret_cast<f32x4*>(out + node * H)[q] = acc;
            }
        }
        // sum over the wavefront with DPP adds (rows of 16, then lane 15 / 31 of the rows below: the total is lane 63's), read
        // back through a scalar register -- no LDS permutes; a fixed order
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < D) {
                float v = part[k];
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, false));
                part[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
            }
        }
        if (lane < D) {
            float total = 0.0f, mine = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k == lane) { total = part[k]; mine = ci[k]; }
            }
            if (mean_coords && deg > 0) total = total / (float)deg;
            coord_out[node * D + lane] = mine + total;
        }
    }
}

template <int H, int PREC, int MODE, bool ATT = false>
int launch_chain(const ChainArgs& a, int layers, hipStream_t st)
{
    using C = Chain<H, PREC>;
    // ring | biases + first-layer vectors | per-wavefront source ids of the in-kernel aggregation
    // ring | biases + first-layer vectors + scale table | per-wavefront source ids of the in-kernel aggregation | maxima table
    size_t lds = (size_t)kRing * C::CHUNK + sizeof(float) * ((size_t)layers * H + 2 * H + 4 * kScaleSlots) +
                 sizeof(int) * kWaves * 32 + sizeof(uint32_t) * kMaxPositions;
    if (ATT) lds = ((lds + 15) & ~(size_t)15) + sizeof(float) * (H + 4);       // the gate's weight row and bias (par_att)
    static bool granted[64] = {};       // (one flag per instantiation: function-local static of a template)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return MDX_ERR_HIP;
    if (lds > 64 * 1024 && !granted[dev]) {
        if (hipFuncSetAttribute((const void*)egnn_edge_chain_kernel<H, PREC, MODE, ATT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            return MDX_ERR_HIP;
        granted[dev] = true;
    }
    if (lds > 160 * 1024) return MDX_ERR_UNSUPPORTED;
    int64_t tiles = (a.n_edges + kTileEdges - 1) / kTileEdges;
    int cus = 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);       // persistent: one workgroup per CU
    hipLaunchKernelGGL((egnn_edge_chain_kernel<H, PREC, MODE, ATT>), dim3(grid), dim3(kWaves * kWave), lds, st, a);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

}  // namespace

// The ATT = true instantiations of the edge chain, compiled in mdx_egnn_chain_att.hip (which includes this file with
// MDX_CHAIN_ATTENTION_UNIT defined): `args` is a ChainArgs.  Not part of the ABI (hidden visibility).
extern "C" int mdx_chain_launch_attention(const void* args, int hidden, int precision, int layers, mdx_stream_t stream);

#ifdef MDX_CHAIN_ATTENTION_UNIT
extern "C" int mdx_chain_launch_attention(const void* args, int hidden, int precision, int layers, mdx_stream_t stream)
{
    const ChainArgs& a = *static_cast<const ChainArgs*>(args);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MDX_ATT_CASE(HH) \
    case HH: return precision == 0 ? launch_chain<HH, 0, 2, true>(a, layers, st) : (precision == 1 ? launch_chain<HH, 1, 2, true>(a, layers, st) : launch_chain<HH, 2, 2, true>(a, layers, st));
    switch (hidden) {
        MDX_ATT_CASE(32)
        MDX_ATT_CASE(64)
        MDX_ATT_CASE(128)
        MDX_ATT_CASE(256)
    }
#undef MDX_ATT_CASE
    return MDX_ERR_UNSUPPORTED;
}
#endif

#ifndef MDX_CHAIN_NO_ENTRY_POINTS      // (the attention unit and the probe units of tools/ include this file without them)
extern "C" {

int64_t mdx_egnn_chain_image_bytes(int hidden, int n_layers)
{
    if (hidden < 32 || (hidden % 32) || n_layers < 1) return -1;
    return ((int64_t)n_layers * hidden * hidden + 32 * hidden) * 4;      // the layers + the head's 32-row chunk
}

int mdx_egnn_chain_pack(const float* const* weights_host, int n_layers, const float* w_out, int hidden, int precision,
                        uint32_t tied_layers, void* image_out, int32_t* exponents_out, mdx_stream_t stream)
{
    if (!weights_host || !image_out || n_layers < 1 || precision < 0 || precision > 2) return MDX_ERR_INVALID_ARG;
    if (precision >= 1 && !exponents_out) return MDX_ERR_INVALID_ARG;
    if (n_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (hidden != 32 && hidden != 64 && hidden != 128 && hidden != 256) return MDX_ERR_UNSUPPORTED;
    PackArgs a{};
    for (int l = 0; l < n_layers; ++l) {
        if (!weights_host[l]) return MDX_ERR_INVALID_ARG;
        a.w[l] = weights_host[l];
    }
    a.w_out = w_out;
    a.layers = n_layers; a.H = hidden; a.precision = precision; a.image = image_out;
    a.tied = tied_layers; a.exps = exponents_out;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (exponents_out) {
        if (hipMemsetAsync(exponents_out, 0, sizeof(int32_t) * (n_layers + 1), st) != hipSuccess) return MDX_ERR_HIP;
        if (precision >= 1) {
            hipLaunchKernelGGL(egnn_chain_maxabs_kernel, dim3(32, (unsigned)n_layers + 1), dim3(256), 0, st, a);
            hipLaunchKernelGGL(egnn_chain_exponents_kernel, dim3(1), dim3(64), 0, st, a);
        }
    }
    const int64_t total = (int64_t)n_layers * hidden * hidden + 32 * hidden;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(egnn_chain_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_edge_chain(const mdx_egnn_chain_t* c, const float* node_proj, const float* coord, int coord_dimension,
                        const int64_t* edges, int64_t n_edges, const int64_t* n_edges_dev, float* messages_out,
                        float* edge_scalar_out, uint32_t* status, mdx_stream_t stream)
{
    if (!c || n_edges < 0 || coord_dimension < 1) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers < 1 || c->n_coord_layers < 1 || (c->precision < 0 || c->precision > 2) ||
        (c->message_mode != MDX_EGNN_MESSAGES_ROWS && c->message_mode != MDX_EGNN_MESSAGES_PIECE_SUMS))
        return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers + c->n_coord_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (c->hidden != 32 && c->hidden != 64 && c->hidden != 128 && c->hidden != 256) return MDX_ERR_UNSUPPORTED;
    if (n_edges == 0) return MDX_OK;
    if (!c->weight_image || !c->biases || !c->bias_in || !c->w_radial || !node_proj || !coord || !edges ||
        !messages_out || !edge_scalar_out || (c->precision >= 1 && !c->weight_exponents))
        return MDX_ERR_INVALID_ARG;
    ChainArgs a{};
    a.image = (const char*)c->weight_image; a.biases = c->biases; a.bias_in = c->bias_in; a.w_radial = c->w_radial;
    a.exps = c->weight_exponents;
    a.act_exps = c->activation_exponents; a.act_max = c->activation_maxima;
    a.node_proj = node_proj; a.coord = coord; a.edges = edges; a.n_edges_dev = n_edges_dev;
    a.n_edges = n_edges; a.n_message = c->n_message_layers; a.n_coord = c->n_coord_layers; a.D = coord_dimension;
    a.messages = messages_out; a.edge_scalar = edge_scalar_out; a.status = status;
    a.piece_sums = c->message_mode == MDX_EGNN_MESSAGES_PIECE_SUMS;
    if ((c->attention_weight == nullptr) != (c->attention_bias == nullptr)) return MDX_ERR_INVALID_ARG;
    a.att_w = c->attention_weight; a.att_b = c->attention_bias;
#ifdef MDX_CHAIN_STAMPS
    // (diagnostic builds: `status` is the caller's stamp list, uint64 [4096]; its entry count restarts with every launch)
    if (status && hipMemsetAsync((char*)status + 4095 * 8, 0, 8, reinterpret_cast<hipStream_t>(stream)) != hipSuccess) return MDX_ERR_HIP;
    a.stamp_buf = status;
    a.status = nullptr;
#endif
    const int layers = a.n_message + a.n_coord;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // The attention instantiations live in a translation unit of their own (mdx_egnn_chain_att.hip: the same source, compiled
    // beside this one): piece-sums mode only -- what E_GCL's forward runs.
    if (a.att_w) return a.piece_sums ? mdx_chain_launch_attention(&a, c->hidden, c->precision, layers, stream) : MDX_ERR_UNSUPPORTED;
#define MDX_CHAIN_MODE(HH, MM, AA) (c->precision == 0 ? launch_chain<HH, 0, MM, AA>(a, layers, st) : (c->precision == 1 ? launch_chain<HH, 1, MM, AA>(a, layers, st) : launch_chain<HH, 2, MM, AA>(a, layers, st)))
#define MDX_CHAIN_CASE(HH)                                                                                            \
    case HH:                                                                                                          \
        return a.piece_sums ? MDX_CHAIN_MODE(HH, 2, false) : MDX_CHAIN_MODE(HH, 0, false);
    switch (c->hidden) {
        MDX_CHAIN_CASE(32)
        MDX_CHAIN_CASE(64)
        MDX_CHAIN_CASE(128)
        MDX_CHAIN_CASE(256)
    }
#undef MDX_CHAIN_CASE
#undef MDX_CHAIN_MODE
    return MDX_ERR_UNSUPPORTED;
}

int mdx_egnn_chain_adapt_activation_exponents(uint32_t* maxima_inout, int count, int32_t* exponents_inout, mdx_stream_t stream)
{
    if (!maxima_inout || !exponents_inout || count < 1 || count > MDX_EGNN_CHAIN_MAX_LAYERS + 2) return MDX_ERR_INVALID_ARG;
    hipLaunchKernelGGL(egnn_chain_act_exponents_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), maxima_inout,
                       count, exponents_inout);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int64_t mdx_egnn_piece_rows(int64_t n_edges, int64_t n_nodes)
{
    if (n_edges < 0 || n_nodes < 0) return -1;
    return ((n_edges + 15) >> 4) + n_nodes;
}

int mdx_segment_combine(const float* pieces, int64_t n_edges, const int64_t* offsets, const int64_t* degree, int64_t n_nodes,
                        int H, int mean, const float* left, float* out, mdx_stream_t stream)
{
    if (n_nodes < 0 || H < 4 || n_edges < 0) return MDX_ERR_INVALID_ARG;
    if (H & 3) return MDX_ERR_UNSUPPORTED;
    if (n_nodes == 0) return MDX_OK;
    if (!pieces || !offsets || !degree || !out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_nodes * kWave + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(segment_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       pieces, offsets, degree, n_nodes, H, mean, out, left, (n_edges + 15) >> 4);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_egnn_node_gather(const float* pieces, int64_t n_edges, const int64_t* offsets, const int64_t* degree, int64_t n_nodes,
                         int H, int mean_messages, const float* left, float* out, const float* edge_scalar, const float* coord,
                         int coord_dimension, const int64_t* edges, int mean_coords, int coord_flags, float* coord_out,
                         mdx_stream_t stream)
{
    if (n_nodes < 0 || H < 4 || n_edges < 0 || coord_dimension < 1) return MDX_ERR_INVALID_ARG;
    if (coord_flags & ~(MDX_EGNN_COORD_NORMALIZE | MDX_EGNN_COORD_TANH)) return MDX_ERR_INVALID_ARG;
    if ((H & 3) || coord_dimension > 8) return MDX_ERR_UNSUPPORTED;
    if (n_nodes == 0) return MDX_OK;
    if (!pieces || !offsets || !degree || !out || !edge_scalar || !coord || !edges || !coord_out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_nodes * kWave + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_node_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       pieces, offsets, degree, n_nodes, H, mean_messages, out, left, (n_edges + 15) >> 4, edge_scalar, coord, edges,
                       coord_dimension, mean_coords, coord_flags, coord_out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

int mdx_mlp_chain_rows(const mdx_egnn_chain_t* c, const float* x, const float* residual, int64_t n_rows,
                       const int64_t* n_rows_dev, float* out, uint32_t* status, mdx_stream_t stream)
{
    if (!c || n_rows < 0) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers < 1 || c->n_coord_layers != 0 || (c->precision < 0 || c->precision > 2)) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (c->hidden != 32 && c->hidden != 64 && c->hidden != 128 && c->hidden != 256) return MDX_ERR_UNSUPPORTED;
    if (n_rows == 0) return MDX_OK;
    if (!c->weight_image || !c->biases || !x || !out || (c->precision >= 1 && !c->weight_exponents)) return MDX_ERR_INVALID_ARG;
    ChainArgs a{};
    a.image = (const char*)c->weight_image; a.biases = c->biases; a.exps = c->weight_exponents;
    a.act_exps = c->activation_exponents; a.act_max = c->activation_maxima;
    a.n_edges_dev = n_rows_dev; a.n_edges = n_rows; a.n_message = c->n_message_layers; a.n_coord = 0; a.D = 0;
    a.rows_in = x; a.residual = residual; a.rows_out = out; a.status = status; a.ld_in = c->hidden;
    const int layers = a.n_message;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MDX_ROWS_CASE(HH)                                                                          \
    case HH: return c->precision == 0 ? launch_chain<HH, 0, 1>(a, layers, st) : (c->precision == 1 ? launch_chain<HH, 1, 1>(a, layers, st) : launch_chain<HH, 2, 1>(a, layers, st));
    switch (c->hidden) {
        MDX_ROWS_CASE(32)
        MDX_ROWS_CASE(64)
        MDX_ROWS_CASE(128)
        MDX_ROWS_CASE(256)
    }
#undef MDX_ROWS_CASE
    return MDX_ERR_UNSUPPORTED;
}

// node_in: [n_rows][ld] rows whose first H columns are h; agg: [n_rows][ld_agg]
static int node_mlp_rows(const mdx_egnn_chain_t* c, const float* node_in, int64_t ld, const float* agg, int64_t ld_agg,
                         int add_residual, int64_t n_rows, const int64_t* n_rows_dev, float* out, float* proj_out,
                         uint32_t* status, mdx_stream_t stream);

int mdx_node_mlp_rows(const mdx_egnn_chain_t* c, const float* node_in, int add_residual, int64_t n_rows,
                      const int64_t* n_rows_dev, float* out, float* proj_out, uint32_t* status, mdx_stream_t stream)
{
    if (!c) return MDX_ERR_INVALID_ARG;
    return node_mlp_rows(c, node_in, 2 * (int64_t)c->hidden, node_in ? node_in + c->hidden : nullptr, 2 * (int64_t)c->hidden,
                         add_residual, n_rows, n_rows_dev, out, proj_out, status, stream);
}

int mdx_node_mlp_rows_split(const mdx_egnn_chain_t* c, const float* h, const float* agg, int add_residual, int64_t n_rows,
                            const int64_t* n_rows_dev, float* out, float* proj_out, uint32_t* status, mdx_stream_t stream)
{
    if (!c) return MDX_ERR_INVALID_ARG;
    return node_mlp_rows(c, h, c->hidden, agg, c->hidden, add_residual, n_rows, n_rows_dev, out, proj_out, status, stream);
}

static int node_mlp_rows(const mdx_egnn_chain_t* c, const float* node_in, int64_t ld, const float* agg, int64_t ld_agg,
                         int add_residual, int64_t n_rows, const int64_t* n_rows_dev, float* out, float* proj_out,
                         uint32_t* status, mdx_stream_t stream)
{
    if (!c || n_rows < 0) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers < 3 || c->n_coord_layers != 0 || (c->precision < 0 || c->precision > 2)) return MDX_ERR_INVALID_ARG;
    if (c->n_message_layers > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    if (c->hidden != 32 && c->hidden != 64 && c->hidden != 128 && c->hidden != 256) return MDX_ERR_UNSUPPORTED;
    if (n_rows == 0) return MDX_OK;
    if (!c->weight_image || !c->biases || !node_in || !agg || !out || (c->precision >= 1 && !c->weight_exponents)) return MDX_ERR_INVALID_ARG;
    ChainArgs a{};
    a.image = (const char*)c->weight_image; a.biases = c->biases; a.exps = c->weight_exponents;
    a.act_exps = c->activation_exponents; a.act_max = c->activation_maxima;
    a.n_edges_dev = n_rows_dev; a.n_edges = n_rows; a.n_message = c->n_message_layers; a.n_coord = 0; a.D = 0;
    a.rows_in = node_in; a.residual = add_residual ? node_in : nullptr; a.rows_out = out; a.status = status;
    a.ld_in = ld; a.rows_in2 = agg; a.ld_in2 = ld_agg;
    a.proj_out = proj_out;
    if (proj_out && c->n_message_layers + 2 > MDX_EGNN_CHAIN_MAX_LAYERS) return MDX_ERR_UNSUPPORTED;
    const int layers = a.n_message;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#define MDX_NODE_CASE(HH)                                                                          \
    case HH: return c->precision == 0 ? launch_chain<HH, 0, 3>(a, layers, st) : (c->precision == 1 ? launch_chain<HH, 1, 3>(a, layers, st) : launch_chain<HH, 2, 3>(a, layers, st));
    switch (c->hidden) {
        MDX_NODE_CASE(32)
        MDX_NODE_CASE(64)
        MDX_NODE_CASE(128)
        MDX_NODE_CASE(256)
    }
#undef MDX_NODE_CASE
    return MDX_ERR_UNSUPPORTED;
}

int mdx_egnn_coord_aggregate(const float* edge_scalar, const float* coord, int coord_dimension, const int64_t* edges,
                             const int64_t* offsets, const int64_t* degree, int64_t n_nodes, int mean, int coord_flags,
                             float* coord_out, mdx_stream_t stream)
{
    if (n_nodes < 0 || coord_dimension < 1) return MDX_ERR_INVALID_ARG;
    if (coord_flags & ~(MDX_EGNN_COORD_NORMALIZE | MDX_EGNN_COORD_TANH)) return MDX_ERR_INVALID_ARG;
    if (n_nodes == 0) return MDX_OK;
    if (!edge_scalar || !coord || !edges || !offsets || !degree || !coord_out) return MDX_ERR_INVALID_ARG;
    int64_t blocks = (n_nodes * coord_dimension + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(egnn_coord_aggregate_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       edge_scalar, coord, edges, offsets, degree, n_nodes, coord_dimension, mean, coord_flags, coord_out);
    return hipGetLastError() == hipSuccess ? MDX_OK : MDX_ERR_HIP;
}

}  // extern "C"
#endif
