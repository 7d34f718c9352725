/*
 * sculpt_hip.h -- C ABI of libsculpt_hip.so, the MI355X (gfx950) implementation of the
 * TripoSR generation hot path of SculptMate.
 *
 * Every entry point replaces a piece of the reference's Python/torch hot path (file:line are
 * relative to the reference repository root):
 *
 *   sculpt_triplane_query        TriplaneNeRFRenderer.query_triplane + NeRFMLP.forward
 *                                  TripoSR/tsr/models/nerf_renderer.py:41-91
 *                                  TripoSR/tsr/models/network_utils.py:116-124
 *   sculpt_plane_features +
 *   sculpt_density_grid          TSR.extract_mesh's dense query over MarchingCubeHelper.grid_vertices
 *                                  TripoSR/tsr/system.py:171-183, TripoSR/tsr/models/isosurface.py:25-39
 *   sculpt_mc_*                  MarchingCubeHelper.forward -> skimage.measure.marching_cubes(vol, 0.0)
 *                                  TripoSR/tsr/models/isosurface.py:41-54
 *   sculpt_gemm_bf16 / sculpt_attention_bf16 / sculpt_layernorm / sculpt_groupnorm_tokens /
 *   sculpt_vit_* / sculpt_upsample_scatter
 *                                Transformer1D / BasicTransformerBlock / Attention / GEGLU,
 *                                  HF ViTModel, TriplaneUpsampleNetwork
 *                                  TripoSR/tsr/models/transformer/transformer_1d.py:179-219
 *                                  TripoSR/tsr/models/transformer/basic_transformer_block.py:149-206,291-315
 *                                  TripoSR/tsr/models/transformer/attention.py:569-653
 *                                  TripoSR/tsr/models/tokenizers/image.py:41-60
 *                                  TripoSR/tsr/models/network_utils.py:24-32
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch types.  All data pointers are DEVICE
 *     pointers (HBM) unless the parameter name ends in _host.
 *   - every function returns 0 on success, non-zero on error; the message is available from
 *     sculpt_last_error() (thread local).  Nothing is retained past the call.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Launch functions are
 *     asynchronous and graph-capturable unless stated otherwise.
 *   - there is NO CPU fallback: without a usable HIP device the launch functions fail.
 */
#ifndef SCULPT_HIP_H
#define SCULPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: sculpt_attention_bf16 no longer takes scale == 0 for pre-scaled queries (sculpt_attention_bf16_prescaled); batched
 *    attention, host-side PLY face records; sculpt_gemm_f32 forwards to sculpt_gemm_f32_ex and takes its two preconditions
 *    (bias 16-byte aligned; without a bias at most 65 532 output columns, see there)
 * 3: the two-pass dense density grid (sculpt_density_grid_filtered, sculpt_density_filter_workspace_bytes,
 *    sculpt_density_filter_stats); added without a version change (new symbols only): sculpt_limbs_bytes, sculpt_limbs_split,
 *    sculpt_gemm_l3p, sculpt_layernorm_limbs, sculpt_attention_f32_l3_limbs, sculpt_mc_count_launch, sculpt_mc_count_read,
 *    sculpt_mc_emit_capped, sculpt_attention_f32_l3_batched, sculpt_mc_count_launch_signed,
 *    sculpt_density_filter_sign_offset
 * 4: the two-pass grid's guard: 12 statistics words instead of 8 (sculpt_density_filter_stats fills SCULPT_FILTER_STATS_WORDS),
 *    word 1 covers every re-evaluated point, the audit sample and the sign mismatches are new; the marching-cubes workspace
 *    is a record pool with a capacity (SCULPT_ERR_MC_WORKSPACE, sculpt_mc_workspace_bytes_for, sculpt_mc_count_launch_for,
 *    sculpt_mc_count_read_ex) */
#define SCULPT_ABI_VERSION 4

typedef void *sculpt_stream_t;

int sculpt_version(void);
/* sha256 (first 32 hex digits) of the HIP sources, headers and compile flags this library was built from
 * (sculptmate_amd/build.py: source_digest); "unknown" when built by hand without -DSCULPT_SOURCE_DIGEST. */
const char *sculpt_source_digest(void);
const char *sculpt_last_error(void);
/* number of visible HIP devices (0 if none); never fails */
int sculpt_device_count(void);
/* A HIP stream whose kernels run only on CUs [first_cu, first_cu + n_cus) of the current device (hipExtStreamCreateWithCUMask;
 * the driver stripes the CU numbering over the 8 XCDs, so a contiguous range takes the same share of every XCD): lets the
 * MFMA-bound density grid of image i and the latency-bound transformer of image i+1 run side by side on disjoint CUs
 * (tools/try_cumask.py).  Destroy with sculpt_stream_destroy. */
int sculpt_stream_create_cu_mask(int first_cu, int n_cus, sculpt_stream_t *stream_out);
int sculpt_stream_destroy(sculpt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * NeRF decoder weights.  The reference MLP is Linear(3*C,64)+SiLU, NH x [Linear(64,64)+SiLU],
 * Linear(64,4)  (network_utils.py:48-79; C=40, NH=8 in TripoSR/checkpoints/config.yaml:23-28).
 * sculpt_mlp_pack re-orders the torch-layout weights (HOST pointers, W[l] is [out][in] row
 * major) into the lane order the MFMA kernels read; the caller uploads the packed blob.
 * ------------------------------------------------------------------------------------------ */
size_t sculpt_mlp_packed_bytes(int in_channels, int n_hidden_64);
int sculpt_mlp_pack(const float *const *W_host, const float *const *b_host, int n_layers,
                    const int *dims_host, void *packed_host, size_t packed_bytes);

/* query_triplane at arbitrary points.
 *   planes      f32 [3][C][H][W]   (scene code of one image, TSR.forward output layout)
 *   mlp_packed  blob from sculpt_mlp_pack (device copy); n_hidden_64 = the NH it was packed with
 *   points      f32 [N][3] in (-radius, radius)
 *   outputs (each may be NULL): density [N], features [N][3], density_act [N], color [N][3]
 *   density_act = exp(density + density_bias), color = sigmoid(features)  (nerf_renderer.py:82-87) */
int sculpt_triplane_query(const float *planes, int C, int H, int W, const void *mlp_packed,
                          int n_hidden_64, const float *points, int64_t N, float radius, float density_bias,
                          float *density, float *features, float *density_act, float *color,
                          sculpt_stream_t stream);

/* Same with sampling flags.  SCULPT_QUERY_ALIGN_CORNERS: grid_sample(align_corners=True), as
 * SF3D.query_triplane does (StableFast/sf3d/system.py:170-199) followed by one MaterialMLP head
 * (StableFast/sf3d/models/network.py:148-210) packed into the (density | features) 4-row output layer:
 * a 1-channel head in row 0 (density_act = exp(out + bias) == trunc_exp), a 3-channel head in rows 1..3
 * (color = sigmoid). */
#define SCULPT_QUERY_ALIGN_CORNERS 1u
/* SCULPT_QUERY_CHANNEL_LAST: `planes` is [3][H][W][C] (sculpt_planes_channel_last of the reference layout): one tap
 * is C contiguous floats, ~20x fewer cache lines per point than gathering from C separate channel planes. */
#define SCULPT_QUERY_CHANNEL_LAST 2u
int sculpt_planes_channel_last(const float *planes /* [3][C][H][W] */, int C, int H, int W, float *out /* [3][H][W][C] */,
                               sculpt_stream_t stream);
int sculpt_triplane_query_ex(const float *planes, int C, int H, int W, const void *mlp_packed,
                             int n_hidden_64, const float *points, int64_t N, float radius, float density_bias,
                             unsigned flags, float *density, float *features, float *density_act, float *color,
                             sculpt_stream_t stream);

/* Dense density grid over the lattice slab ix in [x_begin,x_end), flat order ix*R*R + iy*R + iz
 * (isosurface.py:34-37), in two launches:
 *
 * 1. sculpt_plane_features: the lattice is separable, and the first MLP layer is linear in the three
 *    bilinear plane samples, so it is evaluated once per lattice PAIR and plane:
 *      FA[ix][iy] = W0[:, 0:C]   . sample(plane0; x=ix, y=iy) + b0
 *      FB[ix][iz] = W0[:, C:2C]  . sample(plane1; x=ix, z=iz)
 *      FC[iy][iz] = W0[:, 2C:3C] . sample(plane2; y=iy, z=iz)      (nerf_renderer.py:57-68)
 *    axis_coords[R]: coordinate in (-radius,radius) of lattice index i, computed by the host exactly
 *    as the reference does (linspace(0,1,R) then scale_tensor, isosurface.py:28-32, system.py:177-181).
 *    workspace: sculpt_density_grid_workspace_bytes(R, x_end-x_begin) bytes, holds FA|FB|FC.
 * 2. sculpt_density_grid: per point silu(FA+FB+FC) -> hidden layers (fp32 MFMA) -> density row of the
 *    last layer -> out = exp(density + density_bias) + out_add, f32 [(x_end-x_begin)*R*R].
 *    out_add = -threshold gives the volume the reference hands to marching cubes
 *    (system.py:184 + isosurface.py:45).
 * Linearity only regroups the fp32 sum of layer 0 (see DESIGN.md "fused sample+MLP kernel"). */
size_t sculpt_density_grid_workspace_bytes(int R, int nx);
int sculpt_plane_features(const float *planes, int C, int H, int W, const void *mlp_packed,
                          const float *axis_coords, int R, int x_begin, int x_end, float radius,
                          void *workspace, sculpt_stream_t stream);
/* The same tables with flags: SCULPT_QUERY_ALIGN_CORNERS = grid_sample(align_corners=True), SF3D's query (sf3d/system.py:185-195). */
int sculpt_plane_features_ex(const float *planes, int C, int H, int W, const void *mlp_packed,
                             const float *axis_coords, int R, int x_begin, int x_end, float radius, unsigned flags,
                             void *workspace, sculpt_stream_t stream);
int sculpt_density_grid(const void *mlp_packed, int n_hidden_64, int R, int x_begin, int x_end,
                        float density_bias, float out_add, const void *workspace, float *out,
                        sculpt_stream_t stream);
/* Same with flags (values 1 and 2 were the two-limb experiments SCULPT_DENSITY_BF16X3 / _FP16X3 of rounds 2-5: removed, refused).
 * SCULPT_DENSITY_BF16L3: fp32-equivalent mode on the 16-bit matrix pipe (what TSR.extract_meshes uses by default): both
 * operands split EXACTLY into three bf16 limbs (8 + 8 + 8 = 24 significant bits, fp32 exponent range),
 * W.x = W1.x3 + W3.x1 + W2.x2 + W1.x2 + W2.x1 + W1.x1 with every product exact and fp32 accumulation; the three dropped
 * products are below 2^-23 |W||x|.  No range limit, no fallback.  Replaces NeRFMLP's hidden Linear layers,
 * TripoSR/tsr/models/network_utils.py:116-124, inside nerf_renderer.py:82-87. */
#define SCULPT_DENSITY_BF16L3 4u
int sculpt_density_grid_ex(const void *mlp_packed, int n_hidden_64, int R, int x_begin, int x_end,
                           float density_bias, float out_add, const void *workspace, float *out, unsigned flags,
                           sculpt_stream_t stream);
/* The dense grid for marching cubes, filtered (csrc/density_filter.hip): marching cubes (isosurface.py:41-54) reads the magnitude
 * of the volume only at the end points of sign-changing lattice edges and at all corners of the cells whose sign pattern is
 * ambiguous (Lewiner's face / interior tests, centre vertex), and the sign everywhere else, so
 *   pass A evaluates every lattice point with ONE 16-bit product per hidden layer (bf16 operands, or IEEE half with
 *          SCULPT_FILTER_COARSE_FP16; fp32 accumulate) and marks the points with |log(density_act) - log(level)| < margin
 *          (level = -out_add > 0) or a non-finite coarse value,
 *   pass B re-evaluates the marked points with the SCULPT_DENSITY_BF16L3 arithmetic (bit for bit the values
 *          sculpt_density_grid_ex gives there): after it the sign of every lattice point is certain,
 *   pass C lists, from the signs, the values marching cubes reads (minus the marked points) and re-evaluates those.
 * out then holds the exact value wherever marching cubes reads one and a value of the right sign elsewhere, i.e. marching cubes
 * gives the mesh of the full evaluation bit for bit, PROVIDED no coarse error |log d~ - log d| reaches `margin`.  The caller
 * calibrates the margin: SCULPT_FILTER_MARK_ALL marks every point (margin and level unused), so that stats[1] is the largest
 * coarse error over the lattice.  At run time the call reports what it saw of the coarse error (the guard; the caller decides):
 * stats[1] the largest over ALL re-evaluated points, +inf when an UNMARKED point came out on the other side of the level than
 * the coarse pass had it (stats[10] counts those: each proves an error >= margin) or an exact value is a NaN; stats[8] the
 * largest over an audit sample of ~0.5 % of the points nothing else re-evaluates (unmarked, value not read by marching cubes;
 * pseudo-random, fixed per (R, x_begin, x_end)), appended to pass C.  A wrong sign that no mismatch reveals needs a whole
 * 6-connected component of the true inside or outside mis-signed at every one of its points (csrc/density_filter.hip).
 * flags must contain SCULPT_DENSITY_BF16L3; n_hidden_64 >= 1; R <= 1024.
 * filter_workspace: sculpt_density_filter_workspace_bytes(R, x_end - x_begin) bytes; its first SCULPT_FILTER_STATS_WORDS words are
 *   [0] points re-evaluated  [1] float bits of the largest coarse error seen (see above)  [2] points marked in pass A
 *   [3] non-finite coarse values (all marked)  [4] active cells  [5] lattice points  [6] list entries of pass B  [7] of pass C
 *   [8] float bits of the largest coarse error over the audit sample (+inf: an audited sign was wrong)  [9] audit points
 *   [10] unmarked points whose coarse sign was wrong  [11] marked points whose coarse sign pass B corrected
 * valid once the stream has passed this call (sculpt_density_filter_stats copies them to the host and waits). */
#define SCULPT_FILTER_STATS_WORDS 12
#define SCULPT_FILTER_COARSE_FP16 8u
#define SCULPT_FILTER_MARK_ALL 16u
/* run only the named passes (timing the passes one by one: A, then B, then C on the same filter_workspace); none = all three */
#define SCULPT_FILTER_PASS_A 32u
#define SCULPT_FILTER_PASS_B 64u
#define SCULPT_FILTER_PASS_C 128u
size_t sculpt_density_filter_workspace_bytes(int R, int nx);
int sculpt_density_grid_filtered(const void *mlp_packed, int n_hidden_64, int R, int x_begin, int x_end, float density_bias,
                                 float out_add, float margin, const void *workspace, void *filter_workspace, float *out,
                                 unsigned flags, sculpt_stream_t stream);
/* byte offset of the sign planes (uint32 [nx * R][ceil(R / 32)], bit = value > 0 of the final volume) inside the filter workspace */
size_t sculpt_density_filter_sign_offset(int R, int nx);
int sculpt_density_filter_stats(const void *filter_workspace, int32_t *stats /* host, SCULPT_FILTER_STATS_WORDS */, sculpt_stream_t stream);
/* Step 2 for a decoder head with a 3-channel output (SF3D's MaterialMLP heads evaluated on the marching-tetrahedra lattice,
 * sf3d/system.py:141-168 + network.py:148-210): density_act (nullable) = exp(row 0 + density_bias) + out_add as above,
 * features (nullable) f32 [(x_end-x_begin)*R*R][3] = rows 1..3 of the last layer, raw.  Exact-fp32 kernel only. */
int sculpt_grid_decode(const void *mlp_packed, int n_hidden_64, int R, int x_begin, int x_end, float density_bias,
                       float out_add, const void *workspace, float *density_act, float *features, sculpt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Marching cubes (Lewiner), output identical to skimage.measure.marching_cubes(vol, level)
 * with default arguments, including vertex and face ORDER.
 *   vol f32 [n0][n1][n2] C order.  Two phases: count (synchronises the stream, returns sizes),
 *   then emit into caller-allocated buffers.
 *   flags: SCULPT_MC_REFERENCE_ORDER  faces columns reordered [1,0,2] and stored as int64,
 *          verts multiplied by vert_scale (isosurface.py:52-53) then mapped v*a + b
 *          (scale_tensor, system.py:185-189) when SCULPT_MC_AFFINE is set.
 * Errors: SCULPT_ERR_MC_LEVEL (level outside data range -> skimage ValueError),
 *         SCULPT_ERR_MC_EMPTY (no surface -> skimage RuntimeError).
 * ------------------------------------------------------------------------------------------ */
#define SCULPT_MC_FACES_I64 1u
#define SCULPT_MC_REFERENCE_ORDER 2u
#define SCULPT_MC_USE_CLASSIC 4u
/* slab mode (512^3 split along axis 0): SLAB = do not fail on an empty slab;
 * SLAB_HALO_LOW = lattice plane 0 is the previous slab's last plane: its x/y-edge vertices are owned
 * there, faces reference them as -(1 + axis*n1*n2 + i1*n2 + i2) and are resolved after the gather
 * through the previous slab's top_plane_map (int32 [2][n1][n2], -1 = no vertex). */
#define SCULPT_MC_SLAB 8u
#define SCULPT_MC_SLAB_HALO_LOW 16u
/* sculpt_mc_count_launch_signed / sculpt_mc_count_read after it: the count phase took its "value > level" bits from caller-supplied
 * planes; no data range is collected -- an empty surface comes back as SCULPT_ERR_MC_EMPTY whether or not the level lies inside the
 * range (call sculpt_mc_count to tell skimage's two errors apart), minmax is (+FLT_MAX, -FLT_MAX) */
#define SCULPT_MC_SIGNED 32u
#define SCULPT_ERR_MC_LEVEL 11
#define SCULPT_ERR_MC_EMPTY 12
#define SCULPT_ERR_MC_NAN 13 /* the volume contains NaN (e.g. a 16-bit split density mode left its range) */
#define SCULPT_ERR_MC_WORKSPACE 14 /* more active cells than the workspace's record pool holds: repeat with a larger one (below) */

/* Workspace: per-row arrays + 8 bytes per ACTIVE cell (a cell whose corner signs differ) in a record pool.
 * sculpt_mc_workspace_bytes sizes the pool for one active cell per 8 cells (a closed surface at 256^3 has ~1 per 17): 23 MB at
 * 256^3, 181 MB at 512^3; sculpt_mc_workspace_bytes_for for max_active_cells of them (<= 0: the default).  A count phase that runs
 * out of pool still returns the right totals, with SCULPT_ERR_MC_WORKSPACE and -- through sculpt_mc_count_read_ex -- the number
 * of active cells; the caller allocates sculpt_mc_workspace_bytes_for(.., that many) and repeats the count with
 * sculpt_mc_count_launch_for(.., max_active_cells = that many, ..).  (The emit phase after an overflow writes nothing.) */
size_t sculpt_mc_workspace_bytes(int n0, int n1, int n2);
size_t sculpt_mc_workspace_bytes_for(int n0, int n1, int n2, int64_t max_active_cells);
int sculpt_mc_count(const float *vol, int n0, int n1, int n2, double level, unsigned flags,
                    void *workspace, int64_t *n_verts_host, int64_t *n_faces_host,
                    float *minmax_host /* [2] data min,max or NULL */, sculpt_stream_t stream);
int sculpt_mc_emit(const float *vol, int n0, int n1, int n2, double level, unsigned flags,
                   void *workspace, float vert_div, float vert_mul, float vert_add,
                   int axis0_offset /* slab: global index of lattice plane 0 */,
                   float *verts, void *faces, int *top_plane_map /* or NULL */, sculpt_stream_t stream);
/* The same two phases without the host round trip between them (the stream otherwise idles while the host reads the counts,
 * allocates and launches: ~50 us of a 0.28 ms stage at 256^3):
 *   sculpt_mc_count_launch  the count phase, launched only: the totals stay in the workspace header;
 *   sculpt_mc_emit_capped   the emit phase into buffers of cap_verts vertices / cap_faces faces sized by the CALLER'S ESTIMATE
 *                           (e.g. the previous mesh + 25 %); the kernels read the totals on the device and write NOTHING when the
 *                           mesh does not fit either buffer;
 *   sculpt_mc_count_read    synchronises the stream and returns the totals with sculpt_mc_count's error semantics: when they exceed
 *                           the capacities the caller allocates exactly and calls sculpt_mc_emit.
 * sculpt_mc_count == launch + read; sculpt_mc_emit == emit_capped with unbounded capacities. */
int sculpt_mc_count_launch(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace,
                           sculpt_stream_t stream);
/* The count phase when the caller already holds the planes "vol > level" of the whole lattice -- sign_planes[(i0 * n1 + i1) *
 * words_per_row + i2 / 32] bit i2 % 32, bits past n2 zero: what sculpt_density_grid_filtered leaves in its workspace
 * (sculpt_density_filter_sign_offset) for level 0 -- : bricks of cells without a sign change never read the volume (most of it),
 * the others read their rows as before.  Same records, counts, vertices and faces as sculpt_mc_count_launch; pass
 * flags | SCULPT_MC_SIGNED to sculpt_mc_count_read.  Not for slab mode. */
int sculpt_mc_count_launch_signed(const float *vol, const uint32_t *sign_planes, int words_per_row, int n0, int n1, int n2,
                                  double level, unsigned flags, void *workspace, sculpt_stream_t stream);
int sculpt_mc_count_read(int n0, int n1, int n2, double level, unsigned flags, const void *workspace, int64_t *n_verts_host,
                         int64_t *n_faces_host, float *minmax_host /* [2] or NULL */, sculpt_stream_t stream);
/* The count phase into a workspace of sculpt_mc_workspace_bytes_for(n0, n1, n2, max_active_cells) bytes: sign_planes NULL (then
 * words_per_row is ignored) = sculpt_mc_count_launch, non-NULL with SCULPT_MC_SIGNED in flags = sculpt_mc_count_launch_signed.
 * sculpt_mc_count_read_ex: sculpt_mc_count_read + the number of active cells (what a SCULPT_ERR_MC_WORKSPACE caller asks for). */
int sculpt_mc_count_launch_for(const float *vol, const uint32_t *sign_planes /* or NULL */, int words_per_row, int n0, int n1, int n2,
                               double level, unsigned flags, int64_t max_active_cells, void *workspace, sculpt_stream_t stream);
int sculpt_mc_count_read_ex(int n0, int n1, int n2, double level, unsigned flags, const void *workspace, int64_t *n_verts_host,
                            int64_t *n_faces_host, float *minmax_host /* [2] or NULL */, int64_t *n_active_host /* or NULL */,
                            sculpt_stream_t stream);
int sculpt_mc_emit_capped(const float *vol, int n0, int n1, int n2, double level, unsigned flags, void *workspace, float vert_div,
                          float vert_mul, float vert_add, int axis0_offset, float *verts, int64_t cap_verts, void *faces,
                          int64_t cap_faces, int *top_plane_map /* or NULL */, sculpt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Transformer primitives (bf16 storage, fp32 accumulate).  bf16 values are uint16_t bit patterns.
 * ------------------------------------------------------------------------------------------ */
#define SCULPT_EPI_NONE 0
#define SCULPT_EPI_GELU 1   /* out = gelu_erf(acc + bias) */
#define SCULPT_EPI_GEGLU 2  /* W holds [2*N][K]: out[:, n] = (acc_n + b_n) * gelu_erf(acc_{N+n} + b_{N+n}) */
#define SCULPT_EPI_RELU 3   /* out = max(acc + bias, 0)  (SF3D PixelShuffleUpsampleNetwork convs) */

/* out[M][N] = epi(A[M][K] . W[N][K]^T + bias[N]) (+ residual[M][N], fp32)
 *   A, W bf16 (K contiguous); bias fp32 or NULL; residual fp32 [M][ldr] or NULL;
 *   outputs (any subset, at least one): out_f32 [M][ldo] fp32, out_bf16 [M][ldo] bf16,
 *   out_bf16_t [N][ldt] bf16 = the transposed result (used to hand V^T to the attention kernel).
 *   n_split (0 or N = off): columns n < n_split go to out_f32/out_bf16 only, columns n >= n_split go to
 *   out_bf16_t only, at row n - n_split (one launch for a fused [Q|K|V] projection: Q,K token-major, V^T).
 *   lda/ldw/ldo/ldt are row strides in elements.  K % 64 == 0; N % 128 == 0 (GEGLU: W has 2N rows,
 *   N % 64 == 0); any M >= 1 (ragged last tile handled). */
int sculpt_gemm_bf16(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                     const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                     uint16_t *out_bf16_t, int ldt, int n_split, int M, int N, int K, int epilogue,
                     sculpt_stream_t stream);
/* Same with n_store: only output columns n < n_store are written (N stays a multiple of 128, i.e. W and bias are padded;
 * n_store % 4 == 0): lets a layer with few output channels write straight into a narrow slice of a wider activation
 * buffer (`out` pointing at the slice's first column, ldo = the buffer's row stride). */
int sculpt_gemm_bf16_ex(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                        const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                        uint16_t *out_bf16_t, int ldt, int n_split, int n_store, int M, int N, int K, int epilogue,
                        sculpt_stream_t stream);

/* The same GEMM with a LayerNorm folded in (nn.LayerNorm in front of every Linear of the two transformers:
 * basic_transformer_block.py:98,114,132 / HF ViT layernorm_before, layernorm_after).
 *   Producer side (ln->stats_out): besides out_f32 (and its bf16 copy out_bf16) the launch writes, per row, the (mean, M2 =
 *     sum of squared deviations) of every 64-column slice of the fp32 result: stats_out [N/64][stats_ld][2].
 *   Consumer side (ln->stats_in): A holds those UN-normalised rows x in bf16, W = W0 * diag(gamma) (gamma folded in),
 *     bias = bias0 + W0 . beta, colsum[n] = sum_k W[n][k]; the slice statistics are combined per row (parallel-variance
 *     formula) into (mean, rstd) and   out[m][n] = epi(rstd[m] * (acc[m][n] - mean[m] * colsum[n]) + bias[n])
 *     == epi(LayerNorm(x)[m] . W0[n] + bias0[n]) -- no LayerNorm launch, no normalised copy of x in HBM. */
typedef struct sculpt_ln_fold {
    const float *stats_in; /* [slots_in][stats_ld][2] (slice-major: the rows of one slice are contiguous) or NULL */
    int slots_in;          /* K / 64 */
    const float *colsum;   /* [N] (GEGLU: [2N]) */
    float eps;
    float *stats_out;      /* [N/64][stats_ld][2] or NULL */
    int stats_ld;          /* rows per slice plane of stats_in / stats_out, >= M */
    int rows_per_image;    /* 0, or the rows of ONE image when A stacks the token rows of several (a batched pass): the tile form
                            * is then the one a single image takes (on a taller grid), so that every output and statistic is
                            * accumulated in the single-image order -- the batched pass gives each image the bits of its own pass.
                            * The statistics pointers may all be NULL when only this hint is wanted.  (ABI 4) */
} sculpt_ln_fold_t;
int sculpt_gemm_bf16_ln(const uint16_t *A, int lda, const uint16_t *W, int ldw, const float *bias,
                        const float *residual, int ldr, float *out_f32, uint16_t *out_bf16, int ldo,
                        uint16_t *out_bf16_t, int ldt, int n_split, int n_store, int M, int N, int K, int epilogue,
                        const sculpt_ln_fold_t *ln /* or NULL */, sculpt_stream_t stream);
/* (mean, M2) of every 64-column slice of x [rows][cols] fp32 -> stats [cols/64][stats_ld][2], and the bf16 copy of x: the
 * producer side of the fold for rows that do not come out of a GEMM (the ViT's embedding output). cols % 64 == 0. */
int sculpt_row_slice_stats(const float *x, int ldx, int rows, int cols, float *stats, int stats_ld, uint16_t *x_bf16, int ldb,
                           sculpt_stream_t stream);

/* fp32 "parity mode" of the same stack (exact-fp32 MFMA, ~5x slower; not timed by bench.py):
 *   out[m][n] = epi(alpha * A[m][:].W[n][:] + bias[n]) (+ residual); A, W, out fp32; K % 16 == 0; N % 4 == 0
 *   (w_rows = valid rows of W when N is padded; 0 = N); n_split / out_t as in sculpt_gemm_bf16.
 *   Preconditions since ABI 2 (both entries; a violation is refused with an error, never computed wrongly):
 *     - bias, when given, is 16-byte aligned (the epilogue reads it as float4; a view at a 4- or 8-byte offset must be copied);
 *     - without a bias the per-column vector comes from a 256 KiB zero page: N (2 N with the GEGLU epilogue) + 4 <= 65 536. */
int sculpt_gemm_f32(const float *A, int lda, const float *W, int ldw, const float *bias, const float *residual, int ldr,
                    float *out, int ldo, float *out_t, int ldt, int n_split, int w_rows, int M, int N, int K,
                    float alpha, int epilogue, sculpt_stream_t stream);
/* The same GEMM with a choice of arithmetic and a batch dimension (grid z):
 *   arithmetic SCULPT_F32_EXACT  = v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain (sculpt_gemm_f32);
 *              SCULPT_F32_BF16L3 = fp32 arithmetic on the bf16 matrix pipe: A and W are split EXACTLY into three bf16 limbs each
 *                                  while staged (x = x1 + x2 + x3, 24 significant bits, fp32 exponent range), the six limb products
 *                                  of order >= 2^-16 are exact in fp32 and accumulate in fp32 (csrc/gemm_l3.hip); K % 32 == 0.
 *                                  What TSR(precision="bf16l3") runs its Linears and per-head attention products on: the
 *                                  reference's fp32 Linears (TripoSR/tsr/models/transformer/attention.py:569-653,
 *                                  basic_transformer_block.py:291-315) to fp32 rounding.
 *   batch > 1: entry z reads A + z*a_bs, W + z*w_bs and writes out / out_t (+ reads residual) + z*o_bs (element strides,
 *              multiples of 4): the heads of one attention as ONE launch. */
#define SCULPT_F32_EXACT 0
#define SCULPT_F32_BF16L3 1
int sculpt_gemm_f32_ex(const float *A, int lda, const float *W, int ldw, const float *bias, const float *residual, int ldr,
                       float *out, int ldo, float *out_t, int ldt, int n_split, int w_rows, int M, int N, int K, float alpha,
                       int epilogue, int arithmetic, int batch, int64_t a_bs, int64_t w_bs, int64_t o_bs, sculpt_stream_t stream);
/* Fused attention of the fast parity mode: O = softmax(Q K^T * scale) V per head (head dim 64), fp32 in and out, both products
 * with three-limb bf16 operands and fp32 accumulation (SCULPT_F32_BF16L3 arithmetic), the softmax in fp32 registers
 * (csrc/attention_l3.hip; attention.py:629-631 in the reference's own fp32).  Q [Tq][ldq], K [Tk][ldk] with head h at columns
 * 64 h ..; Vt [heads*64][ldvt] = V transposed, ldvt >= round_up(Tk, 64), columns >= Tk finite; O [Tq][ldo]. */
int sculpt_attention_f32_l3(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, float *O, int ldo, int Tq,
                            int Tk, int heads, float scale, sculpt_stream_t stream);
/* "Limbs once" form of the SCULPT_F32_BF16L3 arithmetic (csrc/gemm_l3p.hip): the split of an fp32 value into 16-bit limbs is done
 * ONCE -- weights at load time, activations by the kernel that produces them -- instead of by every column tile of every launch.
 * A limb-tiled matrix X [R][K] (K % 32 == 0) with NL limbs per element is ceil(R / 32) * K * 64 * NL bytes (sculpt_limbs_bytes),
 * 16-byte aligned:
 *     byte offset of limb l (0 = leading) of X[r][k] = (((r / 32) * (K / 8) + k / 8) * NL + l) * 512 + (r % 32) * 16 + (k % 8) * 2
 * (32-row blocks x 8-k chunks x limb: one matrix-instruction fragment = 512 contiguous bytes, a 16-k step of a row block = NL KiB that
 * the GEMM copies into LDS by DMA).  Rows >= R of the last block: zeros from sculpt_limbs_split, unspecified from other producers
 * (they reach only outputs that are never stored).  Formats:
 *   SCULPT_LIMBS_BF16X3  three bf16 limbs, x = x1 + x2 + x3 exactly (24 significant bits, fp32 exponent range); six limb products per
 *                        multiply: the same products in the same order as sculpt_gemm_f32_ex(SCULPT_F32_BF16L3) -- bit-identical.
 *   SCULPT_LIMBS_F16X2   two fp16 limbs: 22 significant bits for |x| >= 2^-3, an absolute error <= 2^-25 below, |x| < 65504 (larger
 *                        values become inf and the result non-finite: the caller checks); three limb products per multiply, each
 *                        exact in fp32 -- half the matrix work.  A weight is stored times a power of two `scale` that puts its largest
 *                        magnitude in [2^14, 2^15) (sculpt_limbs_split's scale; exact) and the GEMM takes alpha = 1 / scale.
 *   sculpt_limbs_split: scale * fp32 [rows][K] (row stride ld, multiple of 4; src and dst 16-byte aligned) -> limb-tiled.
 *   sculpt_gemm_l3p:    out[m][n] = epi(alpha * A[m][:].W[n][:] + bias[n]) (+ residual), A [M][K] and W [N][K] limb-tiled in `format`.
 *                       N % 128 == 0 (GEGLU: N % 64 == 0 and W stored as 32-row blocks in the order value n..n+31, gate n..n+31,
 *                       value n+32.., gate n+32.. per 64 output columns).  Outputs: out / out_t / n_split / residual as
 *                       sculpt_gemm_f32, OR out_lt: the result itself as a limb-tiled [M][N] matrix in out_format (what the next
 *                       Linear reads; excludes the others).
 *   sculpt_layernorm_limbs:        sculpt_layernorm on fp32 rows with the normalised rows written limb-tiled ([rows][cols]; y_f32
 *                                  optional: the same rows in fp32 as well).
 *   sculpt_attention_f32_l3_limbs: sculpt_attention_f32_l3 with O written limb-tiled: query q of this call is row o_row0 + q of a
 *                                  limb-tiled matrix of o_cols (>= heads * 64, multiple of 32) columns, head h at columns 64 h .. */
#define SCULPT_LIMBS_BF16X3 0
#define SCULPT_LIMBS_F16X2 1
size_t sculpt_limbs_bytes(int rows, int K, int format);
int sculpt_layernorm_limbs(const float *x, int ldx, const float *gamma, const float *beta, float eps, void *y_lt, int format,
                           float *y_f32, int ldy, int rows, int cols, sculpt_stream_t stream);
int sculpt_attention_f32_l3_limbs(const float *Q, int ldq, const float *K, int ldk, const float *Vt, int ldvt, void *O_lt, int format,
                                  int o_row0, int o_cols, int Tq, int Tk, int heads, float scale, int two_fp16_limbs,
                                  sculpt_stream_t stream);
/* `batch` fused three-limb attentions of one shape in ONE launch (TSR.forward on a list of images in the tolerance mode): entry b
 * reads Q + b*q_bs, K + b*k_bs, Vt + b*vt_bs (element strides, multiples of 4; vt_bs may be a column offset into one
 * [heads*64][ldvt] array) and writes O + b*o_bs -- or, with O null and O_lt given, rows o_row0 + b*o_row_bs .. of the limb-tiled
 * output of o_cols columns in `format`.
 * two_fp16_limbs != 0 (both entries): both products on TWO fp16 limbs per operand (22 bits; csrc/attention_l2.hip) instead of three
 * bf16 limbs where the pipelined 256-query form runs (the small launches keep three limbs): half the matrix work, the arithmetic
 * of TSR(precision="fp16l2"); the operands of an attention -- scaled queries, keys, values, probabilities -- are O(1). */
int sculpt_attention_f32_l3_batched(const float *Q, int ldq, int64_t q_bs, const float *K, int ldk, int64_t k_bs, const float *Vt,
                                    int ldvt, int64_t vt_bs, float *O, int ldo, int64_t o_bs, void *O_lt, int format, int o_row0,
                                    int o_row_bs, int o_cols, int Tq, int Tk, int heads, int batch, float scale, int two_fp16_limbs,
                                    sculpt_stream_t stream);
int sculpt_limbs_split(const float *src, int ld, int rows, int K, float scale, int format, void *dst, sculpt_stream_t stream);
int sculpt_gemm_l3p(const void *A_lt, const void *W_lt, int format, float alpha, const float *bias, const float *residual, int ldr,
                    float *out, int ldo, float *out_t, int ldt, int n_split, void *out_lt, int out_format, int M, int N, int K,
                    int epilogue, sculpt_stream_t stream);
/* in-place softmax over the first `cols` columns of each row; columns [cols, pad_cols) are set to 0 */
int sculpt_softmax_rows_f32(float *x, int ld, int rows, int cols, int pad_cols, sculpt_stream_t stream);

/* softmax(Q K^T * scale) V per head, no mask (attention.py:629-631; HF ViTSelfAttention).
 *   Q [Tq][ldq], K [Tk][ldk] bf16 with head h at columns h*64 .. h*64+63;
 *   Vt [heads*64][ldvt] bf16 = V transposed (row = h*64 + d, column = key); ldvt >= round_up(Tk, 64)
 *   and the columns >= Tk must hold finite values (they are multiplied by exact zeros);
 *   O [Tq][ldo] bf16.  Head dim is 64.
 *   scale must be positive.  Row extents must stay below 2 GiB (round_up(Tk,128) * ldk * 2 B and 64 * ldvt * 2 B): the K / V
 *   tiles are fetched with 32-bit buffer offsets. */
int sculpt_attention_bf16(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt,
                          int ldvt, uint16_t *O, int ldo, int Tq, int Tk, int heads, float scale,
                          sculpt_stream_t stream);
/* The same with queries that ALREADY carry softmax_scale * log2(e) (folded into the query projection before its bf16
 * rounding): O = softmax_2(Q K^T) V with 2^x in place of e^x; the kernel subtracts the running maximum inside the matrix
 * product and skips one VALU multiply-add per score (1.1-1.2x faster).  What the bf16 transformers use. */
int sculpt_attention_bf16_prescaled(const uint16_t *Q, int ldq, const uint16_t *K, int ldk, const uint16_t *Vt,
                                    int ldvt, uint16_t *O, int ldo, int Tq, int Tk, int heads, sculpt_stream_t stream);

/* `batch` independent attentions of the same shape in ONE launch (TSR.forward on a batch of images,
 * TripoSR/tsr/system.py:82-115: batch_size = rgb_cond.shape[0]; attention.py:629-631 with a leading batch dimension).
 * Entry b reads Q + b*q_bs, K + b*k_bs, Vt + b*vt_bs and writes O + b*o_bs (strides in ELEMENTS, multiples of 8; o_bs of 4):
 * token rows of the entries stacked below each other, V^T either side by side in one [heads*64][ldvt] array (vt_bs = a column
 * offset, (batch-1)*vt_bs + round_up(Tk,64) <= ldvt) or one array per entry (vt_bs >= heads*64*ldvt).  prescaled != 0: queries
 * carry softmax_scale * log2(e) (scale ignored), as sculpt_attention_bf16_prescaled; otherwise scale must be positive. */
int sculpt_attention_bf16_batched(const uint16_t *Q, int ldq, int64_t q_bs, const uint16_t *K, int ldk, int64_t k_bs,
                                  const uint16_t *Vt, int ldvt, int64_t vt_bs, uint16_t *O, int ldo, int64_t o_bs, int Tq, int Tk,
                                  int heads, int batch, int prescaled, float scale, sculpt_stream_t stream);

/* y = LayerNorm(x) * gamma + beta over the last dim (rows x cols), x fp32 or bf16, y bf16 */
int sculpt_layernorm(const float *x_f32, const uint16_t *x_bf16, int ldx, const float *gamma,
                     const float *beta, float eps, uint16_t *y, int ldy, float *y_f32, int rows, int cols,
                     sculpt_stream_t stream);

/* GroupNorm over x [C][T] fp32 (groups of C/G channels x T tokens, transformer_1d.py:183) written
 * transposed as tokens y [T][C] bf16 */
int sculpt_groupnorm_tokens(const float *x, int C, int T, int G, const float *gamma, const float *beta,
                            float eps, uint16_t *y, float *y_f32 /* either output may be NULL */,
                            float *stats_ws /* 2*G floats scratch */, sculpt_stream_t stream);

/* out[c][t] = x[t][c] + residual[c][t]  (proj_out permute + residual, transformer_1d.py:211-217) */
int sculpt_transpose_add(const float *x_tc, const float *residual_ct, float *out_ct, int T, int C,
                         sculpt_stream_t stream);

/* ImagePreprocessor resize (tsr/utils.py:82-88): F.interpolate(bilinear, align_corners=False, antialias=True)
 * on an HWC fp32 image; tmp = Hin*Wout*C floats of scratch. */
int sculpt_resize_aa_bilinear(const float *in_hwc, int Hin, int Win, int C, float *tmp, float *out_hwc, int Hout,
                              int Wout, sculpt_stream_t stream);

/* ViT front end (tokenizers/image.py:48 + HF ViTEmbeddings; DINOv2: sf3d/models/tokenizers/image.py:86 +
 * dinov2.py:176-210): normalise (x-mean)/std and cut the [S][S][3] fp32 image into patch rows
 * [floor(S/P)^2][ld] (the stride-P patch convolution as a GEMM; trailing S - floor(S/P)*P pixels ignored like the
 * convolution does); columns 3*P*P..ld-1 are written as zero (K padding; ld <= 0 means 3*P*P) */
int sculpt_vit_patchify(const float *image_hwc, int S, int P, const float *mean3_host, const float *std3_host,
                        uint16_t *patches, float *patches_f32 /* either may be NULL */, int ld, sculpt_stream_t stream);
/* tokens[0] = cls + pos[0]; tokens[1+i] = patch_out[i] + pos[1+i]  (fp32 residual stream) */
int sculpt_vit_assemble(const float *patch_out, const float *cls, const float *pos, float *tokens,
                        int n_patches, int hidden, sculpt_stream_t stream);

/* ConvTranspose2d(k=2,s=2) epilogue: g [3*S*S][ldg] fp32 (GEMM output, column = co*4 + dy*2 + dx)
 * + bias[Co] -> planes [3][Co][2S][2S]  (network_utils.py:20-32) */
int sculpt_upsample_scatter(const float *g, int ldg, const float *bias, float *planes, int S, int Co,
                            sculpt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * UV-space texture baker (StableFast "next" row).  Replaces the exports of
 * StableFast/sf3d/texture_baker/texture_baker.dll (ABI declared at baker.py:31-57, 91-118; semantics
 * restated in common.py:104-142, 214-229): pixel (x,y) samples (x/res, 1 - y/res); rast = (u,v,w,tri) or
 * (0,0,0,-1); where several triangles cover a sample the LOWEST index wins.
 * ------------------------------------------------------------------------------------------ */
size_t sculpt_bake_workspace_bytes(int res);
/* uv f32 [nv][2], idx i32 [nf][3] (DEVICE), out f32 [res][res][4] */
int sculpt_bake_rasterize(const float *uv, size_t nv, const int *idx, size_t nf, int res, void *workspace,
                          float *out, sculpt_stream_t stream);
/* attr f32 [nv][3], rast f32 [res][res][4] (DEVICE) -> out f32 [res][res][3] */
int sculpt_bake_interpolate(const float *attr, size_t nv, const int *idx, size_t nf, const float *rast, int res,
                            float *out, sculpt_stream_t stream);
/* The DLL's own entry points (HOST pointers, same names/signatures as baker.py declares): */
void rasterize_cpu(const float *uv, size_t nv, const int *idx, size_t nf, long long res, float *out);
void interpolate_cpu(const float *attr, size_t nv, const int *idx, size_t nf, const float *rast, long long res,
                     float *out);

/* StableFast-3D networks (BASELINE config 4).
 *
 * PixelShuffleUpsampleNetwork (StableFast/sf3d/models/network.py:29-75): each 3x3/pad-1 Conv2d is
 *   sculpt_im2col3x3 (channel-last activations [n][S*S][C] -> rows [n*S*S][9*C], k = (ky*3+kx)*C + c,
 *   elements of 2 (bf16) or 4 (f32) bytes) followed by sculpt_gemm_bf16 with W reordered to [Cout][ky][kx][Cin]
 *   (SCULPT_EPI_RELU for all but the last); sculpt_pixel_shuffle = nn.PixelShuffle(r) of the last GEMM's fp32
 *   output [n*S*S][ldg] (column co*r*r + dy*r + dx) into planes [n][Co][S*r][S*r]. */
int sculpt_im2col3x3(const void *in, int n_planes, int S, int C, int elem_bytes, void *out, sculpt_stream_t stream);
int sculpt_pixel_shuffle(const float *g, int ldg, float *planes, int n_planes, int S, int Co, int r,
                         sculpt_stream_t stream);

/* F.normalize(x, dim=-1, p=2, eps) on n rows of 3 (the "normalize_channel_last" head activation, network.py:129-130;
 * may run in place) */
int sculpt_normalize_rows3(const float *x, int64_t n, float eps, float *y, sculpt_stream_t stream);

/* Texture bake, material composition per texel (StableFast/sf3d/system.py:375-440): all inputs are [res*res][3]
 * texel images except rast [res*res][4] (texels with rast[..][3] < 0 are outside every chart and stay 0):
 * albedo = color; bump = clamp(0.5*(pn.t, pn.b, clip(pn.n, 0.3, 1)) + 0.5, 0, 1) with n, t the normalised interpolated
 * normal / tangent, b = normalize(cross(t, n)), pn = normalize(perturb_normal).  bump may be NULL (albedo only). */
int sculpt_bake_material(const float *rast, int res, const float *color, const float *perturb_normal, const float *nrm,
                         const float *tng, float *albedo, float *bump, sculpt_stream_t stream);

/* Stand-in UV layout (NOT the reference's box-projection atlas, which ends in uv_unwrapper.dll): triangle f is drawn
 * isometrically inside cell (f % cols, f / cols) of a cols x rows grid with relative padding; uv f32 [3*nf][2] in
 * face-corner order (what Mesh.unwrap_uv's `uv[indices]` produces, mesh.py:236-262). */
int sculpt_uv_cell_atlas(const float *v_pos, const void *faces, int faces_i64, int64_t nf, int cols, int rows, float padding,
                         float *uv, sculpt_stream_t stream);

/* MarchingTetrahedraHelper (StableFast/sf3d/models/isosurface.py:108-229).
 *   sculpt_mtet_deform: out = grid_vertices + scale * tanh(offsets), scale = (1-0)/resolution (:108-115, 211-216).
 *   Static per-grid tables (host, once): tets i32 [Nt][4]; edges i32 [Ne][2] = the lexicographically sorted unique
 *   undirected edges (a < b) of the grid (== the reference's all_edges, :117-131); tet_edges i32 [Nt][6] = edge
 *   id of each tet's edges in base_tet_edges order (0-1, 0-2, 0-3, 1-2, 1-3, 2-3; :64-69).
 *   sculpt_mtet_count: scans -> number of vertices (sign-changing edges, sdf > 0 is "inside", :143) and faces; keeps
 *   offsets in `workspace` (sculpt_mtet_workspace_bytes).  sculpt_mtet_emit: vertices f32 [Nv][3] =
 *   (pos_a * (-s_b/(s_a-s_b)) + pos_b * (s_a/(s_a-s_b))) * vert_mul + vert_add in the reference's vertex order
 *   (crossing edges sorted), faces i64 [Nf][3] in the reference's order (one-triangle tets, then two-triangle tets).
 *   Bit-identical to the reference given the same pos/sdf. */
int sculpt_mtet_deform(const float *grid_vertices, const float *offsets, int64_t n_vertices, float scale, float *out,
                       sculpt_stream_t stream);
size_t sculpt_mtet_workspace_bytes(int64_t n_edges, int64_t n_tets);
int sculpt_mtet_count(const float *sdf, const int32_t *tets, int64_t n_tets, const int32_t *edges, int64_t n_edges,
                      void *workspace, int64_t *n_verts_host, int64_t *n_faces_host, sculpt_stream_t stream);
int sculpt_mtet_emit(const float *pos, const float *sdf, const int32_t *tets, int64_t n_tets, const int32_t *edges,
                     int64_t n_edges, const int32_t *tet_edges, void *workspace, float vert_mul, float vert_add,
                     float *verts, int64_t *faces, sculpt_stream_t stream);

/* Channel-last bf16 conv-net building blocks (U^2-Net background removal; the reference runs it
 * as an opaque ONNX graph: rembg/sessions/u2net.py:16-46).  Activations are [H*W][ld] bf16, `in`/`out` point at the first
 * channel of a slice, C = real channels (multiple of 8).
 *   sculpt_im2col3x3_dilated: rows [H*W][9*C_pad], k = (ky*3+kx)*C_pad + c, taps at (y+(ky-1)*d, x+(kx-1)*d), zero outside
 *     the image and for c >= C -- followed by sculpt_gemm_bf16_ex = Conv2d(3x3, padding=d, dilation=d) (+ folded
 *     BatchNorm, ReLU epilogue).
 *   sculpt_maxpool2x2_ceil = nn.MaxPool2d(2, stride=2, ceil_mode=True); sculpt_upsample_bilinear_* =
 *     F.interpolate(mode="bilinear", align_corners=False) to (H, W); sculpt_add_bf16 = elementwise sum of two slices;
 *   sculpt_fuse_sigmoid: out[i] = sigmoid(sum_k w[k]*maps[k][i] + bias) (the 1x1 fusion of the six side outputs). */
int sculpt_im2col3x3_dilated(const uint16_t *in, int ld_in, int H, int W, int C, int C_pad, int dilation, uint16_t *out,
                             sculpt_stream_t stream);
/* The same convolution WITHOUT materialising the im2col rows (implicit GEMM): the activation tile of K-step
 * (tap, 64-channel chunk) is fetched by LDS-DMA straight from the shifted pixels of the channel-last input (a zero page for
 * taps outside the image).  in: [n_images][H*W][ld_in] with at least C_pad readable channels from `in` in every pixel row
 * (channels beyond the real ones must be finite; their weights are zero); Wt bf16 [N][9*C_pad], k = (ky*3+kx)*C_pad + c;
 * outputs as sculpt_gemm_bf16_ex (fp32 and/or bf16, row stride ldo, columns < n_store); epilogue NONE or RELU. */
int sculpt_conv3x3_bf16(const uint16_t *in, int ld_in, int n_images, int H, int W, int C_pad, int dilation,
                        const uint16_t *Wt, const float *bias, float *out_f32, uint16_t *out_bf16, int ldo, int n_store,
                        int N, int epilogue, sculpt_stream_t stream);
int sculpt_maxpool2x2_ceil(const uint16_t *in, int ld_in, int H, int W, int C, uint16_t *out, int ld_out, sculpt_stream_t stream);
int sculpt_upsample_bilinear_bf16(const uint16_t *in, int ld_in, int h, int w, int C, uint16_t *out, int ld_out, int H, int W,
                                  sculpt_stream_t stream);
int sculpt_upsample_bilinear_f32(const float *in, int ld_in, int h, int w, float *out, int H, int W, sculpt_stream_t stream);
int sculpt_add_bf16(const uint16_t *a, int lda, const uint16_t *b, int ldb, uint16_t *out, int ldo, int64_t rows, int C,
                    sculpt_stream_t stream);
int sculpt_fuse_sigmoid(const float *maps, int n_maps, int64_t n, const float *w, float bias, float *out, sculpt_stream_t stream);

/* StableFast-3D estimators (SURVEY.md 8b "SF3D boundary": run_image returns roughness / metallic):
 *   sculpt_resize_bilinear_hwc: F.interpolate(bilinear, align_corners=False, no antialias) of an fp32 [Hin][Win][C] image,
 *     each source pixel first multiplied by mul_hw[Hin][Win] when given -- ClipBasedHeadEstimator's input
 *     rgb_cond * mask_cond resized to 224 (sf3d/system.py:326-329, image_estimator/clip_based_estimator.py:95-100);
 *   sculpt_im2col3x3_strided: rows of a 3x3 / padding 0 / stride s convolution over n_groups channel-last maps
 *     [n_groups][S*S][C] treated as ONE image with n_groups*C channels (channel = g*C + c, the reshape at
 *     global_estimator/multi_head_estimator.py:90-94): out [So*So][9*n_groups*C], k = (ky*3+kx)*n_groups*C + g*C + c,
 *     So = (S-3)/s + 1; elements of 2 (bf16) or 4 (f32) bytes, C*elem_bytes a multiple of 16;
 *   sculpt_col_reduce_f32: out[c] = max (mean = 0) or mean (mean != 0) over the rows of x [rows][ld]
 *     (x.amax / x.mean over the pixels, multi_head_estimator.py:96-101). */
int sculpt_resize_bilinear_hwc(const float *in_hwc, const float *mul_hw /* may be NULL */, int Hin, int Win, int C, float *out_hwc,
                               int Hout, int Wout, sculpt_stream_t stream);
int sculpt_im2col3x3_strided(const void *in, int n_groups, int S, int C, int elem_bytes, int stride, void *out, sculpt_stream_t stream);
int sculpt_col_reduce_f32(const float *x, int ld, int rows, int cols, int mean, float *out, sculpt_stream_t stream);

/* Box-projection UV unwrapping (StableFast/sf3d/uv_unwrapper/unwrap.py:625-697), one entry point per
 * stage; the host side (sculptmate_amd/sf3d/unwrap.py) mirrors Unwrapper.forward.  `stats`: device scratch of
 * sculpt_uv_stats_words() 32-bit words, initialised by sculpt_uv_box_project and carried through the later stages of the
 * same mesh.  faces: int32 or int64 [nf][3] (faces_i64).
 *   sculpt_uv_moments         sums9 (device doubles) = sum of x, y, z, xx, xy, xz, yy, yz, zz over the vertices: the statistics
 *                             behind _align_mesh_with_main_axis (:546-623; exact principal axes instead of the reference's
 *                             randomised torch.pca_lowrank)
 *   sculpt_uv_box_project     rot_pos / rot_nrm = rot (row-major 3x3, host) applied to every vertex; face_uv [nf][3][2] and
 *                             chart [nf] in 0..5 = _box_assign_vertex_to_cube_face (:16-122)
 *   sculpt_uv_chart_tangents  vertex_tangents4 [nv][4] (xyz = _calculate_tangents, :239-305); sums42 (device doubles) [6][7] =
 *                             per chart the sum over its corners of the vertex tangent (3), of the expected tangent (3; with
 *                             the reference's F.normalize(x, -1) scaling, :326-341) and the corner count
 *   sculpt_uv_rotate_charts   face_uv rotated per chart by the angle (cos, sin given per chart) about the chart centre and
 *                             stretched to [0, 1] by the chart's joint min / max (:357-381), in place
 *   sculpt_uv_assign_atlas    assigned [nf]: chart c stays c, moves to the overlap slice c + 6, or to 12 ("remaining") -- the
 *                             contract of assign_faces_uv_to_atlas_index in uv_unwrapper.dll (:124-175), own algorithm
 *                             (UV-space z-buffer of res x res pixels per chart, zbuf = 6*res*res uint64 scratch)
 *   sculpt_uv_place           out_uv [nf][3][2] atlas coordinates (:177-237, 383-527); block_scratch: ceil(nf/256) ints */
size_t sculpt_uv_stats_words(void);
int sculpt_uv_moments(const float *v_pos, size_t nv, double *sums9, sculpt_stream_t stream);
int sculpt_uv_box_project(const float *v_pos, const float *v_nrm, size_t nv, const void *faces, int faces_i64, size_t nf,
                          const float *rot9_host, float *rot_pos, float *rot_nrm, float *face_uv, int *chart, unsigned *stats,
                          sculpt_stream_t stream);
int sculpt_uv_chart_tangents(const float *rot_pos, const float *rot_nrm, size_t nv, const void *faces, int faces_i64, size_t nf,
                             const float *face_uv, const int *chart, float *vertex_tangents4, double *sums42, sculpt_stream_t stream);
int sculpt_uv_rotate_charts(float *face_uv, const int *chart, size_t nf, const float *cos6_host, const float *sin6_host, unsigned *stats,
                            sculpt_stream_t stream);
int sculpt_uv_assign_atlas(const float *rot_pos, const void *faces, int faces_i64, size_t nf, const float *face_uv, const int *chart,
                           int res, unsigned long long *zbuf, int *assigned, sculpt_stream_t stream);
int sculpt_uv_place(const float *face_uv, const int *assigned, size_t nf, double island_padding, unsigned *stats, int *block_scratch,
                    float *out_uv, sculpt_stream_t stream);
/* The DLL's own entry point (HOST pointers, same name / signature as unwrap.py:147-154 declares for uv_unwrapper.dll):
 * vertices [nv][3] (already rotated into the principal frame), indices int64 [nf][3], face_uv [nf][3][2], face_index int64 [nf]
 * in 0..5 -> out int64 [nf] in {c, c + 6, 12} */
void assign_faces_uv_to_atlas_index(const float *vertices, size_t nv, const long long *indices, size_t nf, const float *face_uv,
                                    const long long *face_index, long long *out);

/* StableFast geometry tail:
 *   dilate_fill (sf3d/models/utils.py:96-133): img f32 [3][H][W], mask f32 [H][W]; scratch 8*H*W floats
 *   vertex normals / tangents (sf3d/models/mesh.py:66-139): area-weighted face normal / UV tangent splat
 *   (float atomics: the sum order is not reproducible to the last bit), normalise, Gram-Schmidt. */
int sculpt_dilate_fill(const float *img, const float *mask, int H, int W, int iterations, float *scratch, float *out,
                       sculpt_stream_t stream);
int sculpt_vertex_normals(const float *v_pos, size_t nv, const void *faces, int faces_i64, size_t nf, float *out,
                          sculpt_stream_t stream);
int sculpt_vertex_tangents(const float *v_pos, const float *v_tex, const float *v_nrm, size_t nv, const void *faces,
                           int faces_i64, size_t nf, float *count_scratch, float *out, sculpt_stream_t stream);

/* fp32 -> bf16 (round to nearest even), n elements */
int sculpt_cast_bf16(const float *x, uint16_t *y, int64_t n, sculpt_stream_t stream);

/* Triangle remeshing of the SF3D output mesh: the three gpytoolbox calls of Mesh.triangle_remesh
 * (StableFast/sf3d/models/mesh.py:175-237), on the HOST like the reference's (HOST pointers; no GPU work, no stream).
 *   sculpt_mesh_subdivide      = gpytoolbox.subdivide(v, f, iters=...)      midpoint 1 -> 4 subdivision       (mesh.py:187-191)
 *   sculpt_mesh_decimate       = gpytoolbox.decimate(v, f, face_ratio=...)  shortest-edge collapse to the midpoint with the
 *                                link condition, until <= target_faces faces remain (libigl's default)        (mesh.py:195-199)
 *   sculpt_mesh_remesh_botsch  = gpytoolbox.remesh_botsch(v, f, i, h)       Botsch-Kobbelt isotropic remeshing: split > 4/3 h,
 *                                collapse < 4/5 h, valence flips, tangential relaxation, projection onto the input surface
 *                                (project != 0); h <= 0 = mean edge length of the input; boundary vertices stay  (mesh.py:225-230)
 * V f64 [nv][3], F int32 [nf][3].  The result is an opaque object (its size is not known beforehand): read the counts, copy
 * it out with sculpt_mesh_read (V f64 [num_vertices][3], F int32 [num_faces][3]) and release it with sculpt_mesh_free.
 * gpytoolbox / libigl are absent from the reference tree and the build image: PARITY UNPINNED (csrc/remesh_host.h). */
typedef struct sculpt_host_mesh sculpt_host_mesh_t;
int sculpt_mesh_subdivide(const double *V, size_t nv, const int32_t *F, size_t nf, int iters, sculpt_host_mesh_t **out);
int sculpt_mesh_decimate(const double *V, size_t nv, const int32_t *F, size_t nf, size_t target_faces,
                         sculpt_host_mesh_t **out);
int sculpt_mesh_remesh_botsch(const double *V, size_t nv, const int32_t *F, size_t nf, int iters, double h, int project,
                              sculpt_host_mesh_t **out);
size_t sculpt_mesh_num_vertices(const sculpt_host_mesh_t *m);
size_t sculpt_mesh_num_faces(const sculpt_host_mesh_t *m);
int sculpt_mesh_read(const sculpt_host_mesh_t *m, double *V, int32_t *F);
void sculpt_mesh_free(sculpt_host_mesh_t *m);

/* Mesh hand-off, HOST side (no GPU work): the face block of a binary little-endian PLY file -- per face `uchar 3` followed by
 * three int32 -- from the int64 faces TSR.run returns (the reference's `t_pos_idx.cpu().numpy()`, TripoSR/tsr/system.py:200).
 * faces_host int64 [n][3] -> records_host uint8 [n][13].  Thread-safe; callers split n over threads (sculptmate_amd/meshio.py). */
int sculpt_ply_face_records(const int64_t *faces_host, size_t n, uint8_t *records_host);

#ifdef __cplusplus
}
#endif
#endif /* SCULPT_HIP_H */
