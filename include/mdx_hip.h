/*
 * mdx_hip.h -- C ABI of the MI355X (gfx950) sampling hot path.
 *
 * Drop-in boundary for the per-step work of the reference's predictor-corrector sampler
 * (mila-iqia/diffusion_for_multi_scale_molecular_dynamics).  The reference is pure PyTorch and has no FFI of
 * its own; every entry point below names the reference function (file:line, relative to
 * src/diffusion_for_multi_scale_molecular_dynamics/) whose arithmetic it replaces, and INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add at that call site.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless its name ends in _host.
 *   - the caller (PyTorch) owns every buffer; the library never allocates, frees or retains a pointer.
 *   - stateless, re-entrant, stream-ordered: work is enqueued on `stream` (a hipStream_t passed as void*),
 *     no hidden synchronisation, no globals.  Safe to capture into a hipGraph.
 *   - return value: MDX_OK or a negative status; no exceptions or aborts cross the ABI.  Conditions the
 *     reference checks with device-side asserts (cutoff too large, MASK left at the last step) are reported
 *     through an optional device status word (`status`, OR-ed bits below) that the host reads when it chooses.
 *   - layouts: A int64 [B,N]; X float32 [B,N,d]; L float32 [B,d(d+1)/2]; logits float32 [B,N,C];
 *     C = num_atom_types + 1, class C-1 is MASK.  All arrays C-contiguous.
 *   - arithmetic: IEEE binary32 in the exact operation order documented in DESIGN.md ("MDX arithmetic");
 *     integer outputs are bit-identical to oracle/mdx_oracle.c on identical inputs.
 */
#ifndef MDX_HIP_H
#define MDX_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDX_ABI_VERSION 14

/* status codes */
#define MDX_OK 0
#define MDX_ERR_INVALID_ARG (-1)
#define MDX_ERR_UNSUPPORTED (-2)
#define MDX_ERR_HIP (-3)

/* bits of the optional device status word */
#define MDX_STATUS_CUTOFF_TOO_LARGE 1u   /* utils/neighbors.py:107-113 assert               */
#define MDX_STATUS_MASK_AT_LAST_STEP 2u  /* generators/langevin_generator.py:616-620 assert */
#define MDX_STATUS_EGNN_F16_RANGE 4u     /* mdx_egnn_edge_chain, split-f16 mode: an activation left the f16 range     */
#define MDX_STATUS_GRAPH_CAPACITY 8u     /* mdx_radius_graph_fill_capped: more edges than the caller's capacity      */

#define MDX_MAX_CLASSES 8      /* C supported by the fused atom-type kernels */
#define MDX_PREDICTOR 0
#define MDX_CORRECTOR 1

/* tags of the device-RNG specification ("Device RNG") */
#define MDX_TAG_COORD 0
#define MDX_TAG_GUMBEL 1
#define MDX_TAG_LATTICE 2
#define MDX_TAG_INIT 3
#define MDX_TAG_REPAINT_X0 4
#define MDX_TAG_BINARY 5
#define MDX_TAG_REPAINT_Z 6
#define MDX_TAG_REPAINT_U 7
#define MDX_TAG_INIT_LATTICE 8
#define MDX_TAG_RESAMPLE_Z 9
#define MDX_TAG_RESAMPLE_U 10

#if defined(__GNUC__)
#define MDX_API __attribute__((visibility("default")))
#else
#define MDX_API
#endif

typedef void* mdx_stream_t;

/* Device-resident schedule tables (S1), uploaded/built once per generator.
 * Mirrors the Noise / LangevinDynamics namedtuples of noise_schedulers/noise_scheduler.py:348-378. */
typedef struct mdx_schedule {
    int32_t total_time_steps;       /* T */
    int32_t num_classes;            /* C */
    double sigma_min;               /* NoiseParameters.sigma_min as the Python double it is */
    const float* time;              /* [T] */
    const float* sigma;             /* [T] */
    const float* g;                 /* [T] */
    const float* g_squared;         /* [T] */
    const float* epsilon;           /* [T] */
    const float* q_matrix;          /* [T,C,C] */
    const float* q_bar_matrix;      /* [T,C,C] */
    const float* q_bar_tm1_matrix;  /* [T,C,C] */
} mdx_schedule_t;

/* Counter-based RNG request: used only where the corresponding pre-drawn noise pointer is NULL.
 * value(item, k) comes from Philox4x32-10 with key = seed and
 * counter = (item, (call << 8) | k/4, draw, tag); draw = index * draw_stride + draw_offset. */
typedef struct mdx_rng {
    uint64_t seed;
    uint32_t call;          /* index of the sample() call (sub-batch) */
    uint32_t draw_stride;   /* number_of_corrector_steps + 1 */
    uint32_t draw_offset;   /* 0 predictor, 1+m for corrector m */
    uint32_t reserved;      /* 0 */
    const uint32_t* call_dev;   /* nullable: device word that holds the call index instead of `call` -- a launch captured into a
                                   hipGraph then serves every later sample() call of the same shape (the word is rewritten
                                   between replays, like the step index) */
} mdx_rng_t;

/* Per-step flags of PredictorCorrectorSamplingParameters (generators/predictor_corrector_axl_generator.py:22-30). */
typedef struct mdx_pc_flags {
    int32_t atom_type_greedy_sampling;
    int32_t one_atom_type_transition_per_step;
    int32_t use_fixed_lattice_parameters;
    int32_t update_atom_types;      /* predictor: 1; corrector: atom_type_transition_in_corrector */
    float small_epsilon;
} mdx_pc_flags_t;

MDX_API int mdx_abi_version(void);
MDX_API const char* mdx_status_string(int status);

/* S1 -- NoiseScheduler.__init__ (noise_schedulers/noise_scheduler.py:112-267; sigma_calculator.py:72-74,102-104).
 * schedule_type 0 = exponential, 1 = linear.  Outputs: 9 vectors [T] and 3 tensors [T,C,C]. */
MDX_API int mdx_noise_schedule_build(int total_time_steps, int schedule_type, double time_delta, double sigma_min,
                             double sigma_max, double corrector_step_epsilon, int num_classes, float* time,
                             float* sigma, float* sigma_squared, float* g, float* g_squared, float* epsilon,
                             float* sqrt_2_epsilon, float* beta, float* alpha_bar, float* q_matrix,
                             float* q_bar_matrix, float* q_bar_tm1_matrix, mdx_stream_t stream);

/* Step index kept on the device so that one captured graph serves every iteration. */
MDX_API int mdx_index_set(int32_t* d_index, int32_t value, mdx_stream_t stream);
MDX_API int mdx_index_add(int32_t* d_index, int32_t delta, mdx_stream_t stream);

/* Score-network inputs TIME / NOISE as [B,1] tensors -- LangevinGenerator._get_model_predictions
 * (generators/langevin_generator.py:140-143) with the scalars of predictor_step (:559-563) or
 * corrector_step (:719-729).  index = d_index ? *d_index + index_i : index_i. */
MDX_API int mdx_fill_time_sigma(const mdx_schedule_t* sched_host, int mode, int index_i, const int32_t* d_index,
                        float* time_out, float* sigma_out, int64_t batch, mdx_stream_t stream);

/* P1 -- LangevinGenerator._relative_coordinates_update (generators/langevin_generator.py:155-201) +
 * map_relative_coordinates_to_unit_cell (utils/basis_transformations.py:95-119):
 * out = wrap((x + (score_weight*s)/sigma) + noise_weight*z), elementwise over `count` floats. */
MDX_API int mdx_relative_coordinates_update(const float* x, const float* sigma_normalized_scores, const float* z,
                                    float score_weight, float gaussian_noise_weight, float sigma, int64_t count,
                                    float* out, mdx_stream_t stream);

/* P3 -- LangevinGenerator._lattice_parameters_update (generators/langevin_generator.py:485-490). */
MDX_API int mdx_lattice_parameters_update(const float* l, const float* sigma_normalized_scores, const float* z,
                                  float score_weight, float gaussian_noise_weight, float sigma_n, int64_t count,
                                  float* out, mdx_stream_t stream);

/* P1 / P3 with the three scalars read on the device: weights = {score_weight, gaussian_noise_weight, sigma (sigma_n)}.
 * AdaptiveCorrectorGenerator (generators/adaptive_corrector.py:97-148): its step size eps_i is a batch statistic of the
 * scores and the noise -- computed on the device and never read by the host. */
MDX_API int mdx_relative_coordinates_update_dev(const float* x, const float* sigma_normalized_scores, const float* z,
                                        const float* weights, int64_t count, float* out, mdx_stream_t stream);
MDX_API int mdx_lattice_parameters_update_dev(const float* l, const float* sigma_normalized_scores, const float* z,
                                      const float* weights, int64_t count, float* out, mdx_stream_t stream);

/* P2 -- LangevinGenerator._atom_types_update (generators/langevin_generator.py:247-439) with
 * get_probability_at_previous_time_step / get_probability_from_logits (utils/d3pm_utils.py:64-150).
 * q, q_bar, q_bar_tm1: [C,C] of the current step.  gumbel [B,N,C]; u [B,N] (read only if greedy).
 * probabilities_out (nullable) [B,N,C]: the transition probabilities after the greedy adjustment. */
MDX_API int mdx_atom_types_update(const float* logits, const int64_t* atom_types, const float* q, const float* q_bar,
                          const float* q_bar_tm1, const float* gumbel, const float* u, int64_t batch,
                          int number_of_atoms, int num_classes, float small_epsilon, int greedy,
                          int one_transition, int64_t* atom_types_out, float* probabilities_out,
                          mdx_stream_t stream);

/* Fused per-step update: P2 + P1 + P3 of LangevinGenerator.predictor_step (:536-645) or P1 + P3 (+P2) of
 * corrector_step (:693-805) in ONE launch, scalars taken from the device-resident tables.
 * Pre-drawn noise (reference-RNG parity mode): z_coordinates [B,N,d], gumbel [B,N,C], u [B,N], z_lattice [B,nl];
 * any NULL pointer switches that draw to the device RNG (rng).  Outputs may alias inputs.
 * status (nullable): MDX_STATUS_MASK_AT_LAST_STEP is OR-ed in when index 1 leaves a MASK. */
MDX_API int mdx_pc_step_update(const mdx_schedule_t* sched_host, int mode, int index_i, const int32_t* d_index,
                       const mdx_pc_flags_t* flags_host, const int64_t* atom_types, const float* x, const float* l,
                       const float* logits, const float* score_x, const float* score_l, const float* z_coordinates,
                       const float* gumbel, const float* u, const float* z_lattice, mdx_rng_t rng, int64_t batch,
                       int number_of_atoms, int spatial_dimension, int64_t* atom_types_out, float* x_out,
                       float* l_out, uint32_t* status, mdx_stream_t stream);

/* F1 -- RelativeCoordinatesNoiser.get_noisy_relative_coordinates_sample
 * (noisers/relative_coordinates_noiser.py:33-67): out = wrap(x0 + sigma*z). */
MDX_API int mdx_noise_relative_coordinates(const float* x0, const float* z, float sigma, int64_t count, float* out,
                                   mdx_stream_t stream);

/* F2 -- AtomTypesNoiser.get_noisy_atom_types_sample (noisers/atom_types_noiser.py:30-60) with
 * compute_q_at_given_a0 (utils/d3pm_utils.py:23-39): a_t = argmax_c(log(Qbar[a0][c]) - log(-log u_c)). */
MDX_API int mdx_noise_atom_types(const int64_t* a0, const float* q_bar, const float* u, int64_t n_atoms, int num_classes,
                         int64_t* out, mdx_stream_t stream);

/* F1 / F2 / F3 with the reference's own operands -- one sigma per ELEMENT, one cumulative transition matrix per ATOM (the training
 * transform hands the noisers tensors broadcast from per-structure values, data/diffusion/noising_transform.py:140-195):
 *   mdx_noise_relative_coordinates_sigmas   out = wrap(x0 + sigmas * z)           (relative_coordinates_noiser.py:56-66)
 *   mdx_noise_atom_types_per_atom           q_bar [n_atoms, C, C]; a one-hot a0 times q_bar is the row pick a0 -> q_bar[a0,:]
 *                                           (every other term an exact zero)      (atom_types_noiser.py:42-60)
 *   mdx_noise_lattice_parameters            out = sigmas_n * z + l0               (lattice_noiser.py:69-81) */
MDX_API int mdx_noise_relative_coordinates_sigmas(const float* x0, const float* z, const float* sigmas, int64_t count,
                                                  float* out, mdx_stream_t stream);
MDX_API int mdx_noise_atom_types_per_atom(const int64_t* a0, const float* q_bar, const float* u, int64_t n_atoms,
                                          int num_classes, int64_t* out, mdx_stream_t stream);
MDX_API int mdx_noise_lattice_parameters(const float* l0, const float* z, const float* sigmas_n, int64_t count, float* out,
                                         mdx_stream_t stream);

/* R1 -- ConstrainedLangevinGenerator._repaint_composition (generators/constrained_langevin_generator.py:136-163):
 * for each sample and each constrained row k: forward-noise the known (x_k, a_k) to time index `index_i`
 * (NoisingTransform.transform_given_time_index, data/diffusion/noising_transform.py:98-200; identity when
 * index_i == 0) and overwrite row constrained_indices[k] of X and A in place.
 * z [B,N,d] and u [B,N,C] are the FULL-size draws the reference makes (only constrained rows are read);
 * NULL switches to the device RNG. */
MDX_API int mdx_repaint_constrained_rows(const mdx_schedule_t* sched_host, int index_i, const int32_t* d_index,
                                 const float* constrained_x, const int64_t* constrained_a,
                                 const int64_t* constrained_indices, int number_of_constraints, const float* z,
                                 const float* u, mdx_rng_t rng, int64_t batch, int number_of_atoms,
                                 int spatial_dimension, float* x_inout, int64_t* a_inout, mdx_stream_t stream);

/* RePaint resampling ("2000 steps with resampling", BASELINE configs[4]) -- no reference counterpart: the reference's
 * ConstrainedLangevinGenerator (generators/constrained_langevin_generator.py:94-163) has no resampling loop, so this
 * entry is the build-only option `repaint_resampling_steps` of SURVEY section 8(d); with 0 resampling steps it is never
 * called and the reference's behaviour is unchanged.  One step of the FORWARD process from time index i to i+1
 * (i = *d_index + index_i, 1 <= i < T), in place, for every atom of every structure, with the table row idx = i that
 * the predictor step i+1 -> i reads:  X <- wrap(X + g[idx] * z)        (variance-exploding kernel, F1 arithmetic)
 *                                     A <- argmax_c(log Q[idx][A][c] - log(-log u_c))   (one-step D3PM kernel, F2)
 * z [B,N,d] / u [B,N,C] pre-drawn, or NULL for the device RNG (tags MDX_TAG_RESAMPLE_Z / _U). */
MDX_API int mdx_forward_diffusion_step(const mdx_schedule_t* sched_host, int index_i, const int32_t* d_index,
                                       const float* z, const float* u, mdx_rng_t rng, int64_t batch,
                                       int number_of_atoms, int spatial_dimension, float* x_inout, int64_t* a_inout,
                                       mdx_stream_t stream);

/* N1 -- get_periodic_adjacency_information (utils/neighbors.py:36-224) and the EGNN consumer
 * get_edges_with_radial_cutoff (models/egnn_utils.py:107-144).  Two-call protocol:
 *   1. mdx_radius_graph_count  -> counts[B*N] (edges per source atom)
 *   2. caller: offsets = exclusive_scan(counts), E = sum(counts); allocates outputs
 *   3. mdx_radius_graph_fill   -> edges, ordered by (structure, source, destination[, image])
 * unique = 1: one edge per (src,dst) pair whatever the image (torch.unique(dim=1) of the reference), edges
 *             [E,2] int64 with batch-global node indices (b*N + i); image_out unused.
 * unique = 0: one edge per (src,dst,image); edges [E,2] with per-structure indices, image_out[E] in 0..26
 *             (itertools.product(-1,0,1) order, utils/lattice_utils.py:10-29), shifts_out [E,3] (nullable).
 * Positions [B,N,3] and cells [B,3,3].  The reference takes spatial_dimension 1, 2 or 3: a caller with fewer dimensions
 * embeds its problem -- zero coordinates and orthogonal cell vectors of 4 x cutoff in the missing dimensions: no periodic
 * image along them is within the cutoff, the cutoff-too-large status is decided by the real vectors -- and keeps the first
 * d components of the shifts (what the Python host side does: utils/neighbors.py::embed_in_three_dimensions). */
MDX_API int mdx_radius_graph_count(const float* cartesian_positions, const float* basis_vectors, float radial_cutoff,
                           int64_t batch, int number_of_atoms, int unique, int64_t* counts, uint32_t* status,
                           mdx_stream_t stream);
MDX_API int mdx_radius_graph_fill(const float* cartesian_positions, const float* basis_vectors, float radial_cutoff,
                          int64_t batch, int number_of_atoms, int unique, const int64_t* offsets,
                          int64_t* edges_out, int32_t* image_out, float* shifts_out, mdx_stream_t stream);
/* The same with a caller-sized edge list ("capacity + overflow flag"): step 2 happens on the device --
 * offsets = exclusive scan of counts, E = its last element + last count, both left in device memory -- and the outputs
 * hold `capacity` rows, so no host read sits between count and fill (the step can be captured into a hipGraph).  Edges
 * beyond `capacity` are not written and MDX_STATUS_GRAPH_CAPACITY is OR-ed into `status` (nullable).  capacity =
 * batch * N * (N - 1) can never overflow in unique mode. */
MDX_API int mdx_radius_graph_fill_capped(const float* cartesian_positions, const float* basis_vectors, float radial_cutoff,
                                         int64_t batch, int number_of_atoms, int unique, const int64_t* offsets,
                                         int64_t capacity, int64_t* edges_out, int32_t* image_out, float* shifts_out,
                                         uint32_t* status, mdx_stream_t stream);

/* The graph EGNNScoreNetwork builds per forward (models/score_networks/egnn_score_network.py:236-247): the unique-pair radius
 * graph of RELATIVE coordinates [batch, N, 3] in the orthogonal cell diag(max(lattice_parameters[b, 0..2], clip_min)) (the
 * reference clips the cell lengths to 2.2 x cutoff and zeroes the angles); no host read, nothing but the caller's buffers
 * (counts, offsets [batch*N]; n_edges [1]; edges_out [capacity, 2]).  Cartesian positions are formed in the kernels as
 * relative x length, the bits of the reference's matmul with the diagonal cell.  lattice_stride = row length of
 * lattice_parameters (>= 3).  Status bits as mdx_radius_graph_count / _fill_capped.
 * With a `workspace` of mdx_egnn_radius_graph_workspace_words(batch, N) 64-bit words (caller-owned, need not be initialised, not
 * read after the call) the build is TWO launches and every pair is tested once: the first keeps each source row's hits as 64-bit
 * words in the workspace, the second sums the totals of the structures in front, scans the row counts and writes the pairs.
 * The helper returns 0 where that form does not apply (N > 1024 or batch > 2048); then, or with workspace = NULL, the build is
 * count -> device-side exclusive scan -> fill (three launches).  Same outputs, bit for bit, either way. */
MDX_API int64_t mdx_egnn_radius_graph_workspace_words(int64_t batch, int number_of_atoms);
MDX_API int mdx_egnn_radius_graph(const float* relative_coordinates, const float* lattice_parameters, int lattice_stride,
                                  float clip_min, float radial_cutoff, int64_t batch, int number_of_atoms, int64_t capacity,
                                  int64_t* counts, int64_t* offsets, int64_t* n_edges, int64_t* edges_out, uint32_t* status,
                                  uint64_t* workspace, int64_t workspace_words, mdx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused score network: the reference's MLPScoreNetwork (models/score_networks/mlp_score_network.py:54-370,
 * unconditional forward, no permutation symmetrisation, no time prefactor) evaluated inside the kernels.
 * All pointers are float32 device memory; every w_*_t is the TRANSPOSE of the nn.Linear weight, i.e. [in, out]
 * row-major, so that consecutive lanes (output neurons) read consecutive addresses. */
#define MDX_MLP_MAX_HIDDEN 8
typedef struct mdx_mlp {
    int32_t number_of_atoms, spatial_dimension, num_classes;
    int32_t hidden_size, n_hidden;                       /* hidden_dimensions_size, n_hidden_dimensions */
    int32_t e_coordinates, e_noise, e_time, e_atom_type, e_lattice;   /* embedding sizes */
    const float *w_coordinates_t, *b_coordinates;        /* [2*N*d, e_coordinates] */
    const float *w_noise_t, *b_noise;                    /* [1, e_noise] */
    const float *w_time_t, *b_time;                      /* [1, e_time] */
    const float *w_atom_type_t, *b_atom_type;            /* [C, e_atom_type] */
    const float *w_lattice_t, *b_lattice;                /* [d(d+1)/2, e_lattice] */
    const float* w_hidden_t[MDX_MLP_MAX_HIDDEN];         /* [in_k, hidden_size] */
    const float* b_hidden[MDX_MLP_MAX_HIDDEN];
    const float *w_out_a_t, *b_out_a;                    /* [hidden_size, N*C] */
    const float *w_out_x_t, *b_out_x;                    /* [hidden_size, N*d] */
    const float *w_out_l_t, *b_out_l;                    /* [hidden_size, d(d+1)/2] */
    const float* packed_image;                           /* optional (NULL allowed): all of the above re-laid out by
                                                            mdx_mlp_pack_image; lets a kernel stage the weights into
                                                            LDS with one coalesced copy */
    const float* folded_input;                           /* optional (NULL allowed): the five embedding layers folded
                                                            into hidden layer 0 (no activation lies between them), for the
                                                            input vector [cos 2 pi x (N d) | sin 2 pi x (N d) | sigma | t |
                                                            atom-type embeddings (N e_atom_type) | lattice embedding
                                                            (e_lattice)] of length F: ceil(F/4) x hidden_size weight
                                                            quads [q][neuron][4] (zero-padded), then the folded bias
                                                            [hidden_size].  Used by mdx_mlp_pc_sample for the template
                                                            network; same function, last-bit different rounding */
    const float* folded_output;                          /* optional, used together with folded_input: the last hidden
                                                            layer (no activation follows it) folded into the three heads:
                                                            ceil(hidden_size/4) x (N C + N d + d(d+1)/2) weight quads
                                                            [q][output][4] (outputs: logits | score_x | score_l), then the
                                                            folded bias */
    const float* folded_padded;                          /* optional (ABI v12): the folded layers zero-padded to fixed sizes,
                                                            for the register-resident family of mdx_mlp_pc_sample (hidden_size
                                                            <= 64, N <= 8, <= 64 outputs, F <= 192 folded inputs, 2 .. 4 hidden
                                                            layers): [FQ][64][4] + bias [64] with FQ = 16 ceil(F / 64) quads of
                                                            the folded first layer (neurons >= hidden_size and inputs >= F
                                                            zero) | per middle hidden layer [16][64][4] + bias [64] |
                                                            [16][64][4] + bias [64] of the folded output layer (outputs >= N C
                                                            + N d + d(d+1)/2 zero) */
} mdx_mlp_t;

/* Size (in floats) and construction of the packed weight image referenced by mdx_mlp_t.packed_image. */
MDX_API int64_t mdx_mlp_image_floats(const mdx_mlp_t* mlp_host);
MDX_API int mdx_mlp_pack_image(const mdx_mlp_t* mlp_host, float* image_out, mdx_stream_t stream);

/* ScoreNetwork.forward of the MLP (mlp_score_network.py:281-370 + score_network.py:183-185: MASK logit = -inf):
 * one wavefront per structure, weights streamed from L2, activations in LDS.  time, sigma: [B,1]. */
MDX_API int mdx_mlp_forward(const mdx_mlp_t* mlp_host, const int64_t* atom_types, const float* x, const float* l,
                            const float* time, const float* sigma, int64_t batch, float* logits_out,
                            float* score_x_out, float* score_l_out, mdx_stream_t stream);

/* The whole sampling loop of LangevinGenerator.sample_from_noisy_composition
 * (generators/predictor_corrector_axl_generator.py:145-160 with langevin_generator.py:536-805) in ONE launch for an
 * MLP score network: every wavefront owns one structure, keeps its composition in LDS, and runs
 * `n_iterations` x (predictor + M correctors) -- network forward and fused update -- without returning to the host.
 * Device RNG only (the Philox specification makes it equal, draw for draw, to the per-step kernels).
 * The first iteration's predictor has time index `start_index`; the composition is updated in place.
 * noise_workspace (nullable, device, caller-owned, `workspace_floats` floats): when given, the draws of the segment
 * are generated ahead of the loop by a chip-filling pre-pass kernel into this buffer (same Philox counters, same
 * arithmetic => the same bits) and the persistent kernel only reads them; a workspace smaller than
 * mdx_mlp_pc_sample_workspace_floats(...) splits the segment into several launches.  NULL: every wavefront draws
 * in-kernel.
 * options: OR of the MDX_MLP_SAMPLE_* bits below (0 = the product path).  There are no environment variables: every
 * switch of this entry point is an explicit argument that the caller can print.
 * Record layout of the workspace (what MDX_MLP_SAMPLE_CALLER_NOISE expects the caller to have written), float32:
 *   [iteration it = 0 .. n_iterations-1][structure b][ predictor record rec0 | M corrector records rec1 ]
 *   rec0 = [ z N*d | gumbel N*C | u N | table 8 ]  (table[7] = 0: "no per-step posterior table", always valid)
 *   rec1 = rec0 if atom_type_transition_in_corrector else [ z N*d ]. */
#define MDX_MLP_SAMPLE_GENERIC_KERNEL 1u    /* never select a dimension-specialised instantiation                       */
#define MDX_MLP_SAMPLE_UNFOLDED 2u          /* layer-by-layer forward even when folded_input / folded_output are given
                                             * (the generic kernel too uses them, for any dimensions, when the folded first
                                             * layer is smaller than the two it replaces and the matrices fit in LDS)   */
#define MDX_MLP_SAMPLE_CALLER_NOISE 4u      /* noise_workspace already holds the records (parity tests replay the
                                               reference's recorded draws): the pre-pass is skipped; the whole segment
                                               must fit the workspace                                                    */
#define MDX_MLP_SAMPLE_NO_FIXED_SOFTMAX 8u  /* evaluate the clipped softmax per atom (bit-identical; tests)              */
#define MDX_MLP_SAMPLE_NO_P2_TABLE 16u      /* ignore the per-step posterior table of the records (bit-identical; tests) */
#define MDX_MLP_SAMPLE_PADDED_FAMILY 128u   /* prefer the padded register-resident family (any dimensions within its limits)
                                               to the exact-dimension one where both apply (A/B runs, tests)             */
#define MDX_MLP_SAMPLE_DIAG_NO_FORWARD 256u /* timing diagnostics, honoured only by a -DMDX_DIAGNOSTICS build of the     */
#define MDX_MLP_SAMPLE_DIAG_NO_UPDATE 512u  /* library; the release build returns MDX_ERR_UNSUPPORTED for them           */
MDX_API int64_t mdx_mlp_pc_sample_workspace_floats(const mdx_mlp_t* mlp_host, int number_of_corrector_steps,
                                                   int atom_type_transition_in_corrector, int n_iterations,
                                                   int64_t batch);
/* Which instantiation mdx_mlp_pc_sample runs for this network and these options: 0 = generic (any shape), 1 = the
 * reference template's dimensions as literals (layer by layer), 100 + 10 C + NH = the register-resident family with folded
 * input / output layers: N = 8, d = 3, hidden 64, embeddings 32/16/16/1/1, C in {2,3} classes, NH in {2,3,4} hidden layers
 * (needs folded_input and folded_output); 200 + 10 ceil(F / 64) + NH = the PADDED register-resident family (needs
 * folded_padded: hidden_size <= 64, N <= 8, <= 64 outputs, F <= 192 folded inputs, NH in {2,3,4}): every MLP configuration
 * of the reference's templates and experiments.  -1: invalid descriptor. */
MDX_API int mdx_mlp_pc_sample_variant(const mdx_mlp_t* mlp_host, uint32_t options);
MDX_API int mdx_mlp_pc_sample(const mdx_schedule_t* sched_host, const mdx_mlp_t* mlp_host,
                              const mdx_pc_flags_t* flags_host, int number_of_corrector_steps,
                              int atom_type_transition_in_corrector, int start_index, int n_iterations, mdx_rng_t rng,
                              int64_t batch, int64_t* atom_types, float* x, float* l, float* noise_workspace,
                              int64_t workspace_floats, uint32_t options, uint32_t* status, mdx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * EGNN score network helpers (the forward stays a PyTorch module; these remove passes PyTorch cannot fuse).
 * (No vendor GEMM library behind this ABI: every matrix product of the library is a hand-written MFMA kernel; shapes the
 * edge chain does not cover stay plain PyTorch modules on the caller's side.) */

/* First layer of E_GCL.message_model (models/egnn.py:136-160) on an edge list [E,2] of node indices:
 * out[e,:] = act(node_proj[src_e, :H] + node_proj[dst_e, H:] + bias + radial[e] * w_radial), node_proj [n_nodes, 2H] being
 * the per-node projections h W_src^T | h W_dst^T of the layer's weight.  H % 4 == 0. */
MDX_API int mdx_egnn_message_input(const float* node_proj, const int64_t* edges, const float* radial, const float* bias,
                                   const float* w_radial, int64_t n_edges, int H, int silu, float* out,
                                   mdx_stream_t stream);

/* Segment operations on the radius graph's SORTED edge list: the edges of node i are rows
 * [offsets[i], offsets[i] + degree[i]) (what mdx_radius_graph_fill produces).  One wavefront per node, no atomics,
 * fixed summation order.  H % 4 == 0.
 * mdx_egnn_coord_head: trans[i,:] = (1/degree_i if mean) * sum_e coord_diff[e,:] * (hidden[e,:] . w_out) -- the last
 *   layer of E_GCL.coord_model (nn.Linear(H, 1, bias=False), models/egnn.py:162-200), the product with coord_diff and
 *   unsorted_segment_sum / _mean (models/egnn_utils.py:11-70) in one pass over hidden [E,H]; coord_diff [E,d] with
 *   d <= 64 (the EGNN works on uplifted coordinates, d = 6 for three spatial dimensions).
 * mdx_segment_rows: out[i,:] = (1/degree_i if mean) * sum_e data[e,:] -- the message aggregation of E_GCL.node_model
 *   (models/egnn.py:202-230). */
MDX_API int mdx_egnn_coord_head(const float* hidden, const float* w_out, const float* coord_diff, const int64_t* offsets,
                                const int64_t* degree, int64_t n_nodes, int H, int spatial_dimension, int mean,
                                float* trans, mdx_stream_t stream);
MDX_API int mdx_segment_rows(const float* data, const int64_t* offsets, const int64_t* degree, int64_t n_nodes, int H,
                             int mean, float* out, mdx_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Fused EGNN edge chain on the matrix cores (hand-written MFMA kernel, csrc/mdx_egnn_chain.hip): for every edge of the
 * SORTED edge list, the whole per-edge part of E_GCL.forward (models/egnn.py:136-200, 232-289) in one launch --
 *   x0  = SiLU(node_proj[src,:H] + node_proj[dst,H:] + bias_in + |coord_src - coord_dst|^2 w_radial)   (first message layer)
 *   x   = SiLU(x W_l^T + b_l), l = 0 .. n_message_layers-1                  -> messages_out [E,H]
 *   y   = SiLU(y W_l^T + b_l), l = n_message_layers .. +n_coord_layers-1    (coordinate MLP, all its H -> H layers)
 *   edge_scalar_out[e] = y . w_out                                          (its last layer, Linear(H, 1, bias=False))
 * The [E,H] activations stay in registers between layers (accumulator tile == next layer's MFMA operand).
 * hidden in {32, 64, 128, 256}; message and coordinate MLPs of equal width (a caller with narrower or unequal widths passes
 * zero-padded matrices, biases and vectors: a padded neuron computes SiLU(0) = 0 and feeds zeros on -- what the Python host
 * side does for the reference's default widths 16 / 32); SiLU activations.  E_GCL's options
 * (models/egnn.py:128-135, 148-160, 234-264):
 *   attention  attention_weight [H] / attention_bias [1] (device, both or neither): the messages are gated,
 *              m_e <- m_e sigmoid(m_e . attention_weight + attention_bias), between the message and the coordinate layers --
 *              messages_out and the coordinate MLP see the gated messages; instantiated for message_mode =
 *              MDX_EGNN_MESSAGES_PIECE_SUMS (MDX_ERR_UNSUPPORTED with MDX_EGNN_MESSAGES_ROWS);
 *   tanh, normalize  act where edge_scalar_out meets the coordinate difference: the coord_flags of mdx_egnn_node_gather /
 *              mdx_egnn_coord_aggregate (edge_scalar_out itself is the head's raw value).
 * precision 0: v_mfma_f32_32x32x2_f32, exact binary32 (== fmaf chains in a fixed order).
 * precision 1: split-f16, three v_mfma_f32_32x32x16_f16 per product (hi.hi + hi.lo + lo.hi), binary32 accumulation:
 *              ~2^-22 relative per product.  Both operands are held scaled by exact powers of two so that the f16 halves
 *              keep their 22 bits at small magnitudes: the image of layer l is 2^a_l W_l with the layer's largest |W| in
 *              [2^13, 2^14) (a_l chosen by mdx_egnn_chain_pack and written to weight_exponents), the activations carried
 *              between layers are 2^MDX_EGNN_F16_ACTIVATION_EXPONENT times their value; the kernel undoes both (exact).
 *              Sets MDX_STATUS_EGNN_F16_RANGE when an activation leaves the f16 range: |SiLU output| > ~709.
 * weight_image: the n_message_layers + n_coord_layers matrices [H,H] (nn.Linear layout; weights_host = host array of
 * device pointers) and the head's weight w_out [H], re-laid out by mdx_egnn_chain_pack for the chosen precision
 * (mdx_egnn_chain_image_bytes bytes, caller-owned, 16-byte aligned); the head rides the same pipeline as one more
 * 32-row chunk whose row 0 is w_out.
 * n_edges: number of edge rows to process = capacity of the outputs; n_edges_dev (nullable): device word with the actual
 * count (<= n_edges), for callers that size the edge list without reading it back. */
#define MDX_EGNN_CHAIN_MAX_LAYERS 16
#define MDX_EGNN_MESSAGES_ROWS 0        /* messages_out [E,H] = the messages                                            */
#define MDX_EGNN_MESSAGES_PIECE_SUMS 1  /* messages_out [mdx_egnn_piece_rows(E, n_nodes), H] = per-node piece sums      */
#define MDX_EGNN_F16_ACTIVATION_EXPONENT 6
typedef struct mdx_egnn_chain {
    int32_t hidden, n_message_layers, n_coord_layers, precision;
    int32_t message_mode, reserved;     /* MDX_EGNN_MESSAGES_*; reserved = 0 */
    const void* weight_image;
    const float* biases;       /* [n_message_layers + n_coord_layers, H] */
    const float* bias_in;      /* [H]  bias of the first message layer                       */
    const float* w_radial;     /* [H]  its weight column for the squared distance            */
    const int32_t* weight_exponents;   /* device, [packed layers + 1]: as written by mdx_egnn_chain_pack; required for
                                          precision 1, ignored (may be NULL) for precision 0 */
    /* Split-f16 precisions, nullable (NULL: MDX_EGNN_F16_ACTIVATION_EXPONENT everywhere): one power of two per POSITION of the
     * chain, device, [n_layers + 2], n_layers = n_message_layers + n_coord_layers.  Position q <= n_layers = the operand
     * entering layer q (0: the first layer's output / the rows handed in; n_layers: what leaves the last layer), position
     * n_layers + 1 = the rows read back in front of the projection layers of mdx_node_mlp_rows.  The activations carried at
     * position q are 2^e_q times their value; the f16 range is left at |value| > 65504 / 2^e_q (MDX_STATUS_EGNN_F16_RANGE).
     * mdx_node_mlp_rows treats positions 0 and 1 as one (position 0's exponent).  Read by the kernel at every launch: a
     * launch captured into a hipGraph follows later changes of the array. */
    const int32_t* activation_exponents;
    /* Precision 0, nullable: device, uint32 [n_layers + 2], the float bits of the largest |carried value| seen at each
     * position so far (atomic maxima; the caller zero-fills it) -- the input of mdx_egnn_chain_adapt_activation_exponents. */
    uint32_t* activation_maxima;
    /* E_GCL.att_mlp = Linear(H, 1) + Sigmoid, nullable (both or neither; mdx_egnn_edge_chain only): device, [H] and [1]. */
    const float* attention_weight;
    const float* attention_bias;
} mdx_egnn_chain_t;
#define MDX_EGNN_COORD_NORMALIZE 1   /* coord_diff <- tanh(r^2) / sqrt(r^2 + 1e-16) coord_diff   (E_GCL normalize=True)   */
#define MDX_EGNN_COORD_TANH 2        /* edge_scalar <- tanh(edge_scalar)                          (E_GCL tanh=True)        */
/* exponents_inout[q] = min(exponents_inout[q], e) with 2^e maxima[q] in [2^12, 2^13) (e clamped to [-14,
 * MDX_EGNN_F16_ACTIVATION_EXPONENT]) for every position with a recorded maximum; the maxima are zeroed.  Device-side, no host
 * synchronisation: what a caller runs after an exact-f32 pass that followed an MDX_STATUS_EGNN_F16_RANGE report, so that the
 * split-f16 kernels carry the hot positions with more headroom from then on. */
MDX_API int mdx_egnn_chain_adapt_activation_exponents(uint32_t* maxima_inout, int count, int32_t* exponents_inout,
                                                      mdx_stream_t stream);
MDX_API int64_t mdx_egnn_chain_image_bytes(int hidden, int n_layers);
/* tied_layers: bit l set = layer l uses the same power of two as layer l - 1 (the two H x H halves of a Linear(2H, H), whose
 * accumulators continue one another: mdx_node_mlp_rows).  exponents_out: device, [n_layers + 1] (last: the head row's);
 * required for precision 1; with precision 0 it is zero-filled when given.  No host synchronisation. */
MDX_API int mdx_egnn_chain_pack(const float* const* weights_host, int n_layers, const float* w_out, int hidden,
                                int precision, uint32_t tied_layers, void* image_out, int32_t* exponents_out,
                                mdx_stream_t stream);
MDX_API int mdx_egnn_edge_chain(const mdx_egnn_chain_t* chain_host, const float* node_proj, const float* coord,
                                int coord_dimension, const int64_t* edges, int64_t n_edges, const int64_t* n_edges_dev,
                                float* messages_out, float* edge_scalar_out, uint32_t* status, mdx_stream_t stream);
/* Message aggregation without the [E,H] round trip: with message_mode = MDX_EGNN_MESSAGES_PIECE_SUMS the edge chain adds
 * the messages of a node's edges up INSIDE the kernel, per group of 16 consecutive edge rows (the list is sorted by source),
 * and writes only those sums, into a COMPACT buffer of mdx_egnn_piece_rows(n_edges, n_nodes) = ceil(n_edges / 16) + n_nodes
 * rows (n_edges = the capacity passed to mdx_egnn_edge_chain): the sum over the edges of node i inside group
 * [16 k, 16 k + 16) lands in row k when the group's last edge belongs to node i, otherwise -- the node's last edge is then
 * inside the group -- in the node's own row ceil(n_edges / 16) + i; every other row is left untouched.  No [E,H] buffer
 * exists in this mode (C3: 165 MB instead of 2.1 GB; C5 at 256 structures per GPU: 0.8 GB instead of 12.2 GB).
 * mdx_segment_combine (same n_edges) then gives out[i,:] = (1/degree_i if mean) sum of node i's pieces, read in edge order --
 * unsorted_segment_sum / _mean of the messages (models/egnn_utils.py:11-70) with a fixed summation order and no atomics; it
 * replaces mdx_segment_rows, reads ~2 rows per node instead of degree_i, and the messages themselves never reach memory.
 * Node indices must be < 2^31.
 * left (nullable, [n_nodes,H]): out is [n_nodes, 2H] = [left | sums] -- torch.cat([h, agg], dim=1), the input of the node
 * MLP (models/egnn.py:202-230), written in the same pass. */
MDX_API int64_t mdx_egnn_piece_rows(int64_t n_edges, int64_t n_nodes);
MDX_API int mdx_segment_combine(const float* pieces, int64_t n_edges, const int64_t* offsets, const int64_t* degree,
                                int64_t n_nodes, int H, int mean, const float* left, float* out, mdx_stream_t stream);
/* mdx_segment_combine and mdx_egnn_coord_aggregate (below) as ONE pass over the nodes -- everything E_GCL.forward does per node
 * between the per-edge chain and the node MLP (models/egnn.py:162-230: both unsorted_segment_sum / _mean calls, the product
 * with coord_diff, the coordinate residual, torch.cat([h, agg])): out = [left | message sums] (or the sums alone),
 * coord_out[i,:] = coord[i,:] + (1/degree_i if mean_coords) sum_e (coord[i,:] - coord[dst_e,:]) edge_scalar[e].  The node's
 * edges are dealt to the lanes of one wavefront and reduced by a butterfly: fixed order, no atomics.  coord_dimension <= 8.
 * coord_flags: MDX_EGNN_COORD_* bits (0 for the plain layer). */
MDX_API int mdx_egnn_node_gather(const float* pieces, int64_t n_edges, const int64_t* offsets, const int64_t* degree,
                                 int64_t n_nodes, int H, int mean_messages, const float* left, float* out,
                                 const float* edge_scalar, const float* coord, int coord_dimension, const int64_t* edges,
                                 int mean_coords, int coord_flags, float* coord_out, mdx_stream_t stream);

/* EGNNScoreNetwork's per-node inputs and outputs around the EGNN (models/score_networks/egnn_score_network.py:253-290), one
 * launch each instead of a dozen elementwise passes (spatial dimension 3):
 *   mdx_egnn_node_inputs   z [n_nodes, 2 n_k] = (cos, sin)(2 pi x . K_k) interleaved -- the torus uplift of the relative
 *                          coordinates x [n_nodes,3] with the Bloch wave vectors K [n_k,3] -- and h [n_nodes,H] =
 *                          EGNN.embedding_in([sigma | one_hot(atom type)]) = b + sigma W[:,0] + W[:,1 + a]
 *                          (emb_weight [H, n_features] row-major, n_features = 2 + number of atom types; sigma [B] per
 *                          structure, atoms_per_structure nodes each; the same binary32 operations as the reference's);
 *                          second_out (nullable, [n_nodes, second_width]): a second linear map of the same input,
 *                          b2 + sigma W2[:,0] + W2[:,1 + a] -- with W2 = P W, b2 = P b (formed by the caller) the first
 *                          graph layer's per-node projections P h without an [n_nodes,H] x [H,2H] product;
 *   mdx_egnn_scores        S^alpha = z . Gamma^alpha . x_hat, Gamma^alpha = blockdiag_k(K_k[alpha] [[0,-1],[1,0]]):
 *                          scores [n_nodes,3] from the EGNN's coordinate output x_hat [n_nodes, 2 n_k]. */
MDX_API int mdx_egnn_node_inputs(const float* x, const float* k_vectors, int n_k, const float* sigma, int atoms_per_structure,
                                 const int64_t* atom_types, const float* emb_weight, const float* emb_bias, int n_features,
                                 int H, int64_t n_nodes, float* z_out, float* h_out, const float* second_weight,
                                 const float* second_bias, int second_width, float* second_out, mdx_stream_t stream);
MDX_API int mdx_egnn_scores(const float* z, const float* x_hat, const float* k_vectors, int n_k, int64_t n_nodes,
                            float* scores_out, mdx_stream_t stream);

/* Everything behind the last graph layer in one launch: logits [n_nodes, num_classes] = EGNN.node_classification_layer(h)
 * (models/egnn.py:362-385; class_weight [num_classes, H] row-major, num_classes <= 8) with the logit of class `mask_class` set to
 * -inf (score_network.py:183-185; -1: none), the scores of mdx_egnn_scores, and n_zero zeros into zero_out (nullable with
 * n_zero = 0): the network's all-zero lattice output. */
MDX_API int mdx_egnn_outputs(const float* z, const float* x_hat, const float* k_vectors, int n_k, const float* h,
                             const float* class_weight, const float* class_bias, int H, int num_classes, int mask_class,
                             int64_t n_nodes, float* scores_out, float* logits_out, float* zero_out, int64_t n_zero,
                             mdx_stream_t stream);

/* The same pipeline over the ROWS of a matrix (the per-node MLP of an EGNN layer, models/egnn.py:202-230, after its first
 * layer): out[r,:] = residual[r,:] + W_L (SiLU(W_{L-1} ... SiLU(W_1 x[r,:] + b_1) ...)) + b_L -- L = chain->n_message_layers
 * layers of H x H (chain->n_coord_layers must be 0; bias_in / w_radial unused; image packed with w_out = NULL), every layer
 * but the last followed by SiLU; residual nullable; x, residual, out [n_rows, H] row-major; n_rows_dev nullable as above. */
MDX_API int mdx_mlp_chain_rows(const mdx_egnn_chain_t* chain_host, const float* x, const float* residual, int64_t n_rows,
                               const int64_t* n_rows_dev, float* out, uint32_t* status, mdx_stream_t stream);
/* The WHOLE per-node MLP of an EGNN layer (models/egnn.py:202-230) in one launch: node_in [n_rows, 2H] = [h | agg] (what
 * mdx_segment_combine writes with `left`), first layer Linear(2H, H) + SiLU, then H -> H layers as above, last one linear;
 * out[r,:] = (node_in[r,:H] if add_residual) + MLP(node_in[r,:]).  The chain holds the first layer's weight [H, 2H] as TWO
 * H x H layers -- W[:, :H] then W[:, H:] -- so n_message_layers = 1 + number of Linear modules (>= 3); biases [n, H]: row 0
 * the first layer's bias, row 1 unused.  The first half is multiplied with h, its raw accumulators wait in registers, the
 * second half with agg continues from them.
 * proj_out (nullable, [n_rows, 2H]): the image holds TWO MORE H x H layers behind the n_message_layers of the MLP -- the
 * source and destination halves of the NEXT graph layer's first message layer (models/egnn.py:136-160) -- and
 * proj_out[r,:] = [out[r,:] W_src^T | out[r,:] W_dst^T]: the node_proj input of that layer's mdx_egnn_edge_chain. */
MDX_API int mdx_node_mlp_rows(const mdx_egnn_chain_t* chain_host, const float* node_in, int add_residual, int64_t n_rows,
                              const int64_t* n_rows_dev, float* out, float* proj_out, uint32_t* status, mdx_stream_t stream);
/* The same with the two halves of the row as separate matrices, h [n_rows, H] and agg [n_rows, H] (what mdx_egnn_node_gather
 * writes with left = NULL): the concatenated [h | agg] never exists -- 134 MB less through HBM per layer at C3. */
MDX_API int mdx_node_mlp_rows_split(const mdx_egnn_chain_t* chain_host, const float* h, const float* agg, int add_residual,
                                    int64_t n_rows, const int64_t* n_rows_dev, float* out, float* proj_out, uint32_t* status,
                                    mdx_stream_t stream);
/* coord_out[i,:] = coord[i,:] + (1/degree_i if mean) sum_{e in segment i} (coord[i,:] - coord[dst_e,:]) edge_scalar[e]
 * -- E_GCL.coord_model's trans = coord_diff * coord_mlp(m), unsorted_segment_sum / _mean and the residual add
 * (models/egnn.py:162-200), on the sorted segments; no atomics, fixed summation order. */
MDX_API int mdx_egnn_coord_aggregate(const float* edge_scalar, const float* coord, int coord_dimension, const int64_t* edges,
                                     const int64_t* offsets, const int64_t* degree, int64_t n_nodes, int mean,
                                     int coord_flags, float* coord_out, mdx_stream_t stream);

/* Device-RNG draws as stand-alone fills (trajectory initialisation, tests of the RNG specification).
 * kind 0 = uniform (0,1), 1 = standard normal, 2 = Gumbel(0,1).  out [n_items, width]. */
MDX_API int mdx_rng_fill(int kind, uint64_t seed, uint32_t call, uint32_t draw, uint32_t tag, int64_t n_items, int width,
                 float* out, mdx_stream_t stream);

/* MDX arithmetic probes (tests only): y = f(x) elementwise; fn 0 logf, 1 expf, 2 sinpi, 3 cospi. */
MDX_API int mdx_math_probe(int fn, const float* x, int64_t count, float* y, mdx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MDX_HIP_H */
