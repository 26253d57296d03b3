/*
 * include/lpd_hip.h -- C-ABI of liblpd_hip.so, the MI355X (gfx950) kernels of the LPD-Net
 * global-descriptor hot path.
 *
 * The reference (qiaozhijian/LPD-Net-Pytorch) is pure Python and has no FFI layer of its own; the
 * drop-in boundary is its Python module API (util.PointNetVlad / util.lpdnet_model /
 * loss.pointnetvlad_loss), mirrored by lpd-net-pytorch_amd/.  This header is the boundary UNDER
 * that mirror: each entry point names the reference lines it replaces.  A maintainer binds it with
 * ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C: device pointers, ints, floats; no torch types, no exceptions.
 *   - the caller owns every buffer (inputs, outputs, workspaces); nothing here allocates.  Entry points that reduce over
 *     workgroups (BatchNorm statistics and their backward sums) take `double* stat_ws`: lpd_stat_ws_bytes() bytes of device
 *     memory, zero-filled ONCE by the caller; every call that returns LPD_OK leaves it all-zero again, so one workspace serves
 *     all calls on one stream (two streams that run concurrently need two).  After an LPD_ERR_LAUNCH return re-zero it.
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); nothing synchronises.
 *   - return 0 on success, <0 on error (LPD_ERR_*); lpd_last_error() gives the text
 *     (thread-local).  No global mutable state => callable concurrently from several host
 *     threads (nn.DataParallel-style callers) on different streams/devices.
 *   - all tensors fp32 unless noted; "point-major" = [rows = B*N points][channels], row-major;
 *     "channel-major" = the reference's [B][C][N].
 *   - activation codes: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 sigmoid.
 */
/*
 * Entry points that are NOT on a default path (tools/abi_coverage.py, profiles/r06_abi_coverage.txt: eval forwards and train steps of
 * every trunk in both storage modes and both product modes, ragged cloud sizes, the callers' helpers -- 81 of the 101 entry points
 * are called, 3 more are infrastructure: lpd_version, lpd_last_error, lpd_knn_pm_layout).  The 17 below are SUPERSEDED formulations,
 * kept because tests use them as on-device cross-checks of the fused kernels that replaced them; the Python mirror reaches them only
 * through an LPD_DEBUG token.  A binding that wants the product path does not need them:
 *   lpd_gemm_bf16x1                      one bf16 product per term (LPD_DEBUG=bf16-x1)
 *   lpd_split_panels                     fp32 panels -> split bf16 planes as a pass of its own (the producers write planes)
 *   lpd_edge_gather_max                  K-agg with every gather through L2 (superseded by the cloud-resident / windowed kernels)
 *   lpd_group_max, lpd_group_max_bwd, lpd_scatter_add_rows, lpd_edge_split_fwd (wave-per-point form)
 *                                        materialised [M k, C] edge tensors of rounds 1-2 (the split-form stage works from gather sums)
 *   lpd_edge_build_bf16, lpd_edge_act_max_bf16, lpd_group_sel_stats_bf16, lpd_edge_bn_bwd_bf16, lpd_edge_bn_bwd_bf16_sel,
 *   lpd_gemm_bf16s, lpd_gemm_bf16s_bnbwd, lpd_gather_sum_rows_bf16
 *                                        round 3's chain of passes for the bf16-storage DG1 -> DG2 stage (one launch forward, two backward now)
 *   lpd_gemm_tn_bf16 (+ _ws_floats)      weight gradients on the register-transposing kernel (superseded by the transposed-read kernel)
 * The kNN `impl` values 1 / 2 / 4 / 6 of lpd_knn are test cross-checks and A/B timing forms as well (see lpd_knn below).
 */
#ifndef LPD_HIP_H
#define LPD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPD_OK 0
#define LPD_ERR_ARG (-1)
#define LPD_ERR_LAUNCH (-2)
#define LPD_ERR_UNSUPPORTED (-3)

#define LPD_ACT_NONE 0
#define LPD_ACT_RELU 1
#define LPD_ACT_LEAKY 2
#define LPD_ACT_SIGMOID 3

int lpd_version(void);
const char* lpd_last_error(void);
/* size of the `stat_ws` workspace (see Conventions): 32 replicas x 2 x 1024 doubles = 512 KiB */
long long lpd_stat_ws_bytes(void);

/*
 * kNN graph construction.  Replaces util/lpdnet_model.py:317-326 `knn(x, k)`.
 *   x      [B][C][N] channel-major (exactly the reference's argument layout)
 *   idx    [B][N][k] int32, the k largest pd = -|x_i - x_j|^2 (reference arithmetic order, see
 *          csrc/lpd_knn.hip), descending; self is included; ties -> lower index first.
 *   ws     workspace of lpd_knn_workspace_floats(B, C, N, k) floats: per-point sums of squares, the packed MFMA operand
 *          image xp[b][n][h][s] = x[b][2s+h][n], and (best-first path) per-tile centroids / radii / visiting orders
 *   impl   0 = product path: f32-MFMA distance tiles + queued selection; best-first tile order with exact skip bounds for
 *          C <= 64, k <= 64 (four waves per query tile while the grid would not fill the chip), the first-generation kernel for
 *          64 < C <= 256.  Everything else is NOT a product path: 4 / 6 = ascending scan / best-first walk forced (A/B timing, same
 *          indices), 5 = per-wave walk statistics INSTEAD of indices (tools/knn7_stats.py), 1 = VALU fmaf cross-check of the MFMA
 *          arithmetic (k <= 20; an on-device oracle for tests), 2 = first-generation kernel forced.  Other values: LPD_ERR_ARG.
 *          (impl 3 and the timing ablations 10..137 of rounds 1-2 were removed in round 5.)
 * Supported: C <= 256, k <= 64, k <= N.  Bit-exact vs the reference CPU path on tie-free rows.
 */
int lpd_knn(const float* x, int B, int C, int N, int k, int32_t* idx, float* ws, int impl, void* stream);
long long lpd_knn_workspace_floats(int B, int C, int N, int k);
/* The same on point-major rows x_pm [B*N][ld] (the pipeline's activation layout; C <= 64, k <= 64): no transposes. */
int lpd_knn_pm(const float* x_pm, int ld, int B, int C, int N, int k, int32_t* idx, float* ws, int impl, void* stream);
/* impl | LPD_KNN_PM_PREPARED: the operands of the 64-channel cloud are in ws already (lpd_lpdnet_front wrote them); x_pm may be
 * NULL.  lpd_knn_pm_layout: where lpd_knn_pm keeps them -- squared norms xx [B*N], packed operand image xp [B*N][2][32], the
 * bf16 image xb of the low-precision bound pass (NULL when that pass will not run on these sizes) and the statistics of the
 * 32-point tiles of the best-first search (centroids [B*nt][2 cp] in packed operand order, then |c|^2, radius and max |x|^2,
 * [B*nt] each; nt = ceil(N / 32), cp = 2 for C <= 4 else 32; NULL when the best-first search will not run).  "Prepared" means
 * all of these. */
#define LPD_KNN_PM_PREPARED 256
int lpd_knn_pm_layout(int B, int C, int N, int k, float* ws, float** xx, float** xp, void** xb, float** tiles);

/*
 * The layers in front of the feature-space kNN of LPD-Net in one launch (util/lpdnet_model.py:231-232, no T-Nets):
 *   F0 = act(BN2(conv2_lpd(act(BN1(conv1_lpd(xyz))))))   xyz [B*N][ldx] (3 coordinates used), W1 [64][3], W2 [64][64] ([out][in]),
 *   s / b = folded eval-mode BatchNorm scale / shift, act none / ReLU / LeakyReLU; exact fp32 (fma chains, f32-input MFMA).
 * F0 [B*N][64] row-major.  knn_ws != NULL (a lpd_knn_workspace_floats(B, 64, N, k) workspace): the kNN operands of F0 are
 * written there too, and lpd_knn_pm(NULL, 64, B, 64, N, k, idx, knn_ws, LPD_KNN_PM_PREPARED, stream) builds the graph.
 * N % 128 == 0.
 */
int lpd_lpdnet_front(const float* xyz, int ldx, const float* W1, const float* s1, const float* b1, const float* W2, const float* s2,
                     const float* b2, int act, float slope, float* F0, int B, int N, int k, float* knn_ws, void* stream);

/*
 * Dense fp32 GEMM with fused epilogue:  C = act((A.B + bias) * scale + shift), per output column.
 * Replaces the 1x1 Conv1d/Conv2d + eval BatchNorm + activation stacks
 * (util/lpdnet_model.py:231-232,262,297-305; util/PointNetVlad.py:152-175,213-230), the NetVLAD
 * matmuls (PointNetVlad.py:48,66,76) and the gating matmul (PointNetVlad.py:104).
 *   A logical [M][K]: a_kmajor = 0 -> stored [M][lda]; 1 -> stored [K][lda] (i.e. A^T in memory)
 *   B logical [K][N]: b_kmajor = 1 -> stored [K][ldb]; 0 -> stored [N][ldb] (torch conv weight)
 *   batch > 1: strides sA/sB/sC in elements between problems
 *   splits > 1: split-K; splitk_ws must hold batch*splits*M*N floats; epilogue applied after the sum
 *   bias/scale/shift: [N] or NULL (scale and shift together)
 *   accumulate != 0: C += result (after the epilogue) -- gradient accumulation in the backward pass
 * Any M, N, K (tails are zero-filled / masked).  Requires lda/ldb/sA/sB % 4 == 0, A and B 16-byte aligned.
 *   a_cloud / c_cloud != 0: A / C are in CLOUD-PANEL layout [cloud][cols/8][panel_ld][8] (lda / ldc ignored): rows
 *   m = cloud * panel_n + n, n < panel_n <= panel_ld (the pad rows keep consecutive panels off the same HBM channels);
 *   a_cloud / c_cloud = floats between clouds of that buffer (it may hold more panels than the operand uses: pass the
 *   pointer to the operand's first panel).  An 8-channel slice of one cloud is one contiguous
 *   32*panel_n-byte run -- what lpd_edge_gather_max16 streams -- and a whole cloud one contiguous block.  Needs
 *   a_kmajor = 0, batch = 1, splits = 1, K % 32 == 0, panel_n % 128 == 0.
 */
int lpd_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC, int splits,
             float* splitk_ws, const float* bias, const float* scale, const float* shift, int act, float slope,
             int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream);

/*
 * lpd_gemm on the bf16 MFMA: each fp32 operand is split hi + lo (two bf16) while it is staged into LDS and every
 * product is the sum of three bf16 MFMA products (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi), fp32 accumulation.  Error
 * ~5e-6 of the output range (the f32-input form: ~5e-7); 2e-6 on the final descriptors (DESIGN.md "GEMM precision").
 * Same arguments and layouts as lpd_gemm.
 */
int lpd_gemm_bf16x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC, int splits,
             float* splitk_ws, const float* bias, const float* scale, const float* shift, int act, float slope,
             int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream);

/* The same kernel with ONE product per term (a_hi * b_hi: the operands rounded to bf16, fp32 accumulation): the bf16-storage
 * training mode (BASELINE.json configs[2]).  ~4e-3 relative per operand. */
int lpd_gemm_bf16x1(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
             int a_kmajor, int b_kmajor, int batch, long long sA, long long sB, long long sC, int splits,
             float* splitk_ws, const float* bias, const float* scale, const float* shift, int act, float slope,
             int accumulate, long long a_cloud, long long c_cloud, int panel_n, int panel_ld, void* stream);

/*
 * Split-bf16 GEMM for a weight-shaped B: lpd_gemm_prep_b splits B (either layout) once into hi/lo bf16 stored in MFMA
 * fragment order (lpd_gemm_prep_b_bytes(N, K) bytes, 16-byte aligned); lpd_gemm_x3w then computes
 * C = act((A.B + bias) * scale + shift) for row-major A [M][lda] without staging or splitting B again (each wave streams
 * its 32-column fragments from L2 into registers).  Used for the wide forward layers (conv3 512 -> 1024, the split edge
 * projections) and for dX = dY.W in the backward pass.  a_cloud / c_cloud / panel_n / panel_ld: cloud-panel A / C as in
 * lpd_gemm (0 = row-major).  impl: 0 = by shape (N >= 256: 128 x 256 blocks, each wave a 128 x 64 strip; else 128 x 128),
 * 2 / 3 = 128 x 128 / 128 x 256 blocks forced; impl | 16: one product per term (a_hi * b_hi, see lpd_gemm_bf16x1).
 */
long long lpd_gemm_prep_b_bytes(int N, int K);
int lpd_gemm_prep_b(const float* B, int ldb, int b_kmajor, int N, int K, void* frags, void* stream);
/* `batch` matrices sB floats apart -> `batch` fragment sets lpd_gemm_prep_b_bytes(N, K) bytes apart (lpd_gemm_x3t_rows, batched) */
int lpd_gemm_prep_b_batch(const float* B, int ldb, int b_kmajor, int N, int K, int batch, long long sB, void* frags, void* stream);
int lpd_gemm_x3w(const float* A, int lda, const void* frags, float* C, int ldc, int M, int N, int K, const float* bias,
                 const float* scale, const float* shift, int act, float slope, int accumulate,
                 long long a_cloud, long long c_cloud, int panel_n, int panel_ld, int impl, void* stream);

/* The bare product C = A W^T (+ bias) of a TRAIN-mode layer with its BatchNorm statistics from the epilogue: stat_sum / stat_sumsq
 * [N] fp64 (zeroed by the call) = column sums / sums of squares of C over the M rows (fp32 over a block's 128 rows, fp64 atomics),
 * i.e. lpd_colstats without the second pass over C. */
/* lpd_gemm_x3w with per-problem weights: rows [b batch_rows, (b + 1) batch_rows) of the row-major A take fragment set b of
 * lpd_gemm_prep_b_batch (frag_bytes = lpd_gemm_prep_b_bytes(N, K) apart); batch_rows % 128 == 0.  NetVLAD backward's dA[b] = x[b] . dV[b]. */
/* a_scale / a_shift (or null): the rows of A are act(a_scale[k] A[m][k] + a_shift[k]) -- lpd_gemm_x3w_act's operand transform with
 * nothing stored (bf16 rows: the transformed values rounded to bf16 = the map a bf16-storing lpd_gemm_x3w_act writes) */
int lpd_gemm_x3w_batched(const void* A, int lda, int a_bf16, const void* frags, long long frag_bytes, int batch_rows, float* C, int ldc, int M,
                         int N, int K, const float* a_scale, const float* a_shift, int a_act, float a_slope, int impl, void* stream);
/* bf16 rows as the operand (a_bf16 above, flags & 1 of lpd_gemm_x3w_act, lpd_gemm_x3w_bf16a): A [M][lda] in bf16 elements, row-major,
 * K % 32 == 0.  A plain product takes the rows as the hi image (two MFMA products against the split weight); with an operand
 * transform they are widened first.  flags & 2 of lpd_gemm_x3w_act: a_out is a bf16 tensor, and the product sees the rounded values.
 * The bf16-storage training mode keeps the [B N, 1024] conv3 map, its activated form and its gradient as bf16. */
int lpd_gemm_x3w_bf16a(const void* A16, int lda, const void* frags, float* C, int ldc, int M, int N, int K, int accumulate, int impl, void* stream);
/* c_bf16 & 1: C receives bf16 values ([M][ldc] in bf16 elements, N % 32 == 0); the statistics stay those of the fp32 accumulators.
 * c_bf16 & 2: A holds bf16 rows too (lda in bf16 elements, K % 32 == 0; with a bf16 C and N >= 256): two products per term */
int lpd_gemm_x3w_stats(const float* A, int lda, const void* frags, void* C, int ldc, int c_bf16, int M, int N, int K, const float* bias,
                       double* stat_sum, double* stat_sumsq, int impl, double* stat_ws, void* stream);
/* The product with the train-mode BatchNorm affine + activation of the layer IN FRONT applied in the operand loader:
 * C = act(a_scale[k] A[m][k] + a_shift[k]) W^T (+ bias), the transformed rows stored to a_out [M][a_ld] on the way (or null).
 * Row-major A, N <= 128.  util/lpdnet_model.py:262 (bn3_lpd + act) feeding util/PointNetVlad.py:48 (x . cluster_weights): the
 * stand-alone affine pass over the [B N, 1024] map disappears.  impl: 0, or 16 = plain bf16 operands (bf16 storage mode). */
int lpd_gemm_x3w_act(const void* A, int lda, const void* frags, float* C, int ldc, int M, int N, int K, const float* bias,
                     const float* a_scale, const float* a_shift, int a_act, float a_slope, void* a_out, int a_ld, int flags, int impl, void* stream);

/*
 * The same product for a SHORT reduction with cloud-panel A and C (the neighbour / centre projection of the split SN1 edge
 * convolution, util/lpdnet_model.py:257: K = 128 -> N = 512), computed transposed: the prepared weight fragments are the MFMA's
 * row operand, each wave keeps its 32 data rows (split hi / lo once) in registers for all N columns and stores whole KiB runs of
 * a panel -- no LDS, no barrier.  lpd_gemm_x3t_applies: K in {64, 128}, N % 32 == 0, both operands cloud panels, clouds of a
 * multiple of 128 points, act none / ReLU / LeakyReLU.  frags: lpd_gemm_prep_b(W [N][K], b_kmajor = 0).
 */
int lpd_gemm_x3t_applies(int M, int N, int K, int act, long long a_cloud, long long c_cloud, int panel_n);
/* lpd_gemm_x3t with A given as SPLIT bf16 planes (a_hi: hi plane [cloud][K/8][panel_ld][8], a_cloud elements between clouds; the lo
 * plane a_lo elements behind it) */
int lpd_gemm_x3ts(const void* a_hi, long long a_lo, const void* frags, float* C, int M, int N, int K, const float* bias,
                  const float* scale, const float* shift, int act, float slope, long long a_cloud, long long c_cloud, int panel_n,
                  int panel_ld, void* stream);
/* The transposed short-reduction product on ROW-MAJOR fp32 operands: C [M][ldc] = act((A [M][lda] . W^T + bias) * scale + shift) with the
 * fragments of lpd_gemm_prep_b; K = 64 or 128, N % 32 == 0, M % 128 == 0.  batch > 1: independent problems with their own A, C
 * (strides sA, sC in floats) and fragment sets (frag_bytes apart). */
int lpd_gemm_x3t_rows_applies(int M, int N, int K, int act, long long lda, long long ldc);
/* c_bf16: C receives bf16 values (ldc / sC in bf16 elements, ldc % 8 == 0) */
int lpd_gemm_x3t_rows(const float* A, long long lda, const void* frags, void* C, long long ldc, int c_bf16, int M, int N, int K, const float* bias,
                      const float* scale, const float* shift, int act, float slope, int batch, long long sA, long long sC,
                      long long frag_bytes, void* stream);

int lpd_gemm_x3t(const float* A, const void* frags, float* C, int M, int N, int K, const float* bias, const float* scale,
                 const float* shift, int act, float slope, long long a_cloud, long long c_cloud, int panel_n, int panel_ld,
                 void* stream);

/*
 * kNN-graph aggregation (K-agg).  Replaces the gather/repeat/cat of util/lpdnet_model.py:331-363
 * fused with a split edge convolution + BatchNorm + activation + max over k
 * (lpdnet_model.py:249-250 convDG1/x1, :257-258 convSN1/x3):
 *   out[m][c] = act(scale[c] * (sel_t P[cloud(m)*N + idx[m][t]][c] + Q[m][c]) + shift[c])
 *   sel = max over the k neighbours where scale[c] >= 0, min where scale[c] < 0.
 *   P [M][ldp], Q [M][ldq] or NULL, idx [M][k] (indices local to the cloud), out [M][ldo].
 * C in {64,128,256}; M = B*N; leading dims % 4 == 0; pointers 16-byte aligned.
 */
int lpd_edge_gather_max(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, float* out, int ldo,
                        const float* scale, const float* shift, int M, int N, int C, int k, int act, float slope,
                        void* stream);

/*
 * K-agg, cloud-resident form on row-major operands with 16-bit indices: as lpd_edge_gather_max, but one
 * workgroup keeps an 8-channel slice of ALL N rows of one cloud's P in LDS (N*32 bytes), so the k gathers per
 * point are LDS reads and P, Q, out cross the memory system once.  idx16: the blocked uint16 copy made by lpd_pack_idx16.
 * Built for k = 20, N <= 4096; bit-identical to lpd_edge_gather_max.
 * p_cloud / q_cloud / o_cloud != 0: that operand is in cloud-panel layout [cloud][.][panel_ld][8] (floats between clouds; leading
 * dim ignored): a block's 8-channel slice is then one contiguous 32*N-byte run instead of N pieces of 32 bytes.
 */
int lpd_edge_gather_max16(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, float* out, int ldo,
                          const float* scale, const float* shift, int M, int N, int C, int k, int act, float slope, long long p_cloud,
                          long long q_cloud, long long o_cloud, int panel_ld, void* stream);
/* The same with SPLIT output for lpd_gemm_p8 / lpd_gemm_x3ts: out_hi = hi plane of a pair of bf16 cloud-panel planes
 * [cloud][C/8][panel_ld][8] (hi = bf16(x), lo = bf16(x - hi); o_cloud bf16 elements between clouds), the lo plane o_lo elements
 * behind it.  Same bytes as the fp32 row; the two lanes of a point exchange halves (DPP) so that each still stores 16 bytes. */
int lpd_edge_gather_max16s(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, void* out_hi, long long o_lo,
                           const float* scale, const float* shift, int M, int N, int C, int k, int act, float slope, long long p_cloud,
                           long long q_cloud, long long o_cloud, int panel_ld, void* stream);

/* int32 kNN indices [M][k] (local to the cloud, < 65536) -> the uint16 copy lpd_edge_gather_max16 reads:
 * blocked by 32 points, index quad i of point m at ((m/32)*5 + i)*32 + m%32 (uint2 units); idx16 holds
 * ceil(M/32)*32*k uint16.  k = 20. */
int lpd_pack_idx16(const int32_t* idx, uint16_t* idx16, long long M, int k, void* stream);

/*
 * K-agg for LARGE clouds (N > 4096, e.g. BASELINE configs[4]: N = 16384, k = 64): a Z-order WINDOW of 4095 rows of the cloud's
 * 8-channel slice in LDS, neighbours outside the window from L2 (csrc/lpd_edge_win.hip).  Same result, bit for bit, as
 * lpd_edge_gather_max; operands as lpd_edge_gather_max16 (row-major or cloud-panel).  idx16: the RAW uint16 copy of the
 * indices made by lpd_pack_idx16w (blocked by 32 points, quad i of point m at ((m/32)*(k/4) + i)*32 + m%32, uint2 units;
 * ceil(M/32)*32*k uint16).  k in {20, 32, 64}; N <= 57344; the clouds should be Z-ordered (lpd_morton_sort) for the window to
 * catch most neighbours -- correctness does not depend on it.  lpd_pack_idx16w with N > 0 (points per cloud) also PARTITIONS every
 * list, out-of-window neighbours first (the miss phase then stops at the longest miss list of a wave); N = 0 keeps the kNN order.
 */
int lpd_pack_idx16w(const int32_t* idx, uint16_t* idx16, long long M, int k, int N, void* stream);
int lpd_edge_gather_maxw(const float* P, int ldp, const float* Q, int ldq, const uint16_t* idx16, float* out, int ldo,
                         const float* scale, const float* shift, int M, int N, int C, int k, int act, float slope, long long p_cloud,
                         long long q_cloud, long long o_cloud, int panel_ld, void* stream);

/*
 * Fused per-edge MLP: stage-1 BatchNorm+activation on the fly, stage-2 1x1 conv on the f32 MFMA,
 * BatchNorm + activation + max over k.  Replaces util/lpdnet_model.py:251-252 (convDG2 applied to
 * the un-maxed convDG1 output, then max) and the LPDNetOrign chains lpdnet_model.py:97-100,105-107:
 *   y1[m][t][:] = act(s1 * (P[nbr(m,t)] + Q[m]) + b1)            (never materialised)
 *   out[m][o]   = act(s2[o] * sel_t (W2 y1[m][t])[o] + b2[o])
 *   W2 [CO][CM] torch [out,in] layout.  (CM,CO) in {(128,128),(64,64)}; k <= 128.
 *   out_cloud != 0: out is in cloud-panel layout [cloud][.][panel_ld][8] with out_cloud floats between clouds (ldo ignored).
 */
int lpd_edge_mlp(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                 const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo, int M,
                 int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream);
/* Same contract on the bf16 MFMA (split-bf16, three products per term, fp32 accumulate: see lpd_gemm_bf16x3). */
int lpd_edge_mlp_bf16x3(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                 const float* b1, const float* W2, const float* s2, const float* b2, float* out, int ldo, int M,
                 int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream);
/* ... with SPLIT output (see lpd_edge_gather_max16s): out_hi / out_lo planes of bf16 cloud panels, out_cloud elements between clouds */
int lpd_edge_mlp_bf16x3s(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                 const float* b1, const float* W2, const float* s2, const float* b2, void* out_hi, long long out_lo, int M,
                 int N, int CM, int CO, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream);
/* ... and with x1 = max over k of the stage-1 activation act(s1 (P_j + Q_i) + b1) -- the DG1-stage K-agg of LPDNet.forward
 * (util/lpdnet_model.py:249-250), which is the maximum over the slots of the very tile this kernel builds for the DG2 product --
 * written on the way as a second pair of planes (x1_hi, x1_hi + out_lo; layout of out_hi): one launch instead of lpd_pack_idx16 +
 * lpd_edge_gather_max16s + lpd_edge_mlp_bf16x3s over the same graph.  128 -> 128 channels, M % 32 == 0, N % 64 == 0. */
int lpd_edge_mlp_x1_bf16x3s(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1,
                 const float* b1, const float* W2, const float* s2, const float* b2, void* out_hi, void* x1_hi, long long out_lo,
                 int M, int N, int k, int act, float slope, long long out_cloud, int panel_ld, void* stream);

/* Per-point linear layer with K <= 8 inputs (+bias, affine, activation): the 3 -> 64 first layers
 * (util/lpdnet_model.py:185,231; util/PointNetVlad.py:190,213; T-Net conv1 lpdnet_model.py:276) and the
 * per-cloud 3x3 alignment products x @ trans (lpdnet_model.py:229; PointNetVlad.py:209).
 * Weight element (n, c) of weight set b is W[b*w_sb + n*w_sn + c*w_sk]; row m uses set m / rows_per_w
 * (rows_per_w = 0: one shared weight). */
int lpd_linear_smallk(const float* X, int ldx, const float* W, int w_sn, int w_sk, long long w_sb, int rows_per_w,
                      float* Y, int ldy, int M, int N, int K, const float* bias, const float* scale,
                      const float* shift, int act, float slope, void* stream);

/* Batched transpose in [batch][R][C] -> out [batch][C][R] (point-major <-> channel-major). */
int lpd_transpose(const float* in, float* out, int batch, int R, int C, int ldi, int ldo, long long si,
                  long long so, void* stream);

/* NetVLAD soft-assignment: out[r][:] = softmax(scale * in[r][:] + shift), ncols <= 64
 * (util/PointNetVlad.py:51-58, eval-mode bn1 folded into scale/shift). In-place allowed.
 * colsum != NULL: rows come in groups (clouds) of group_rows (a multiple of 16) and colsum[g * colsum_ld + c] receives
 * (+=, float atomics; the caller zeroes it) the column sums of `out` over group g -- NetVLAD's a_sum (:63) without a
 * second pass over the assignments. */
int lpd_softmax_affine(const float* in, float* out, int rows, int ncols, const float* scale, const float* shift,
                       int group_rows, float* colsum, int colsum_ld, void* stream);

/* NetVLAD residual + normalisations (util/PointNetVlad.py:61-74).
 *   vraw [B][F][KC] = act^T x per cloud, act [B][N][KC], cw2 [F][KC] (cluster_weights2[0]),
 *   out [B][F*KC]: (vraw - a_sum*cw2), L2-normalised over F per cluster, flattened f*KC+c, L2-normalised. KC = 64.
 *   ws: workspace of B*2*KC floats (a_sum and per-cluster sums of squares, zeroed here); asum_ready != 0: ws[b][0..KC)
 *   already holds a_sum (lpd_softmax_affine with colsum = ws, colsum_ld = 2*KC), ws[b][KC..2KC) is zero, act may be NULL.
 *   aux_asum [B][KC], aux_inv_c [B][KC], aux_inv_g [B]: optional (NULL in inference) -- a_sum and the two
 *   reciprocal norms, saved for lpd_vlad_finalize_bwd. */
int lpd_vlad_finalize(const float* vraw, const float* act, const float* cw2, float* out, float* ws, float* aux_asum,
                      float* aux_inv_c, float* aux_inv_g, int B, int N, int F, int KC, int asum_ready, void* stream);

/* Per-cloud max over the N points: in [B][N][ldi] -> out [B][C]
 * (util/PointNetVlad.py:137,162 mp1; util/lpdnet_model.py:300). */
int lpd_colmax(const float* in, int ldi, float* out, int B, int N, int C, void* stream);

/* Context gating in one launch (util/PointNetVlad.py:103-115): out[b][:] = h[b][:] * sigmoid((h[b][:] . Wg + bias) * scale + shift),
 * h [B][ldh], Wg [D][ldw] (gating_weights, k-major), bias / (scale, shift) optional per-column vectors (the eval-mode bn1
 * affine or gating_biases).  out may not alias h. */
int lpd_gating(const float* h, int ldh, const float* Wg, int ldw, const float* bias, const float* scale, const float* shift,
               float* out, int ldo, int B, int D, void* stream);

/* out = a * b elementwise (context gating product, util/PointNetVlad.py:113). */
int lpd_mul(const float* a, const float* b, float* out, long long n, void* stream);

/*
 * Lazy triplet / quadruplet loss, forward + gradient in one launch.  Replaces
 * loss/pointnetvlad_loss.py:6-97 (best_pos_distance, triplet_loss, quadruplet_loss).
 *   q (b,0,d) at q[b*q_sb + d]; pos (b,p,d) at pos[b*pos_sb + p*pos_st + d]; neg likewise; other like q
 *   (strides in elements, unit stride over d) -- the four inputs are normally views of one
 *   [bq, 1+P+Ng+1, D] descriptor tensor (train_pointnetvlad.py:214-217).
 *   quad = 0: triplet form (other/m2/gother ignored).  use_min / lazy / ignore_zero: the reference flags.
 *   loss [1]; minmax [2][bq] (min_pos, max_pos); gradients of the loss: gq [bq][D], gpos [bq][P][D],
 *   gneg [bq][Ng][D], gother [bq][D].
 */
int lpd_metric_loss(const float* q, long long q_sb, const float* pos, long long pos_sb, long long pos_st,
                    const float* neg, long long neg_sb, long long neg_st, const float* other, long long other_sb,
                    int bq, int P, int Ng, int D, float m1, float m2, int use_min, int lazy, int ignore_zero, int quad,
                    float* loss, float* minmax, float* gq, float* gpos, float* gneg, float* gother, void* stream);

/* Backward of best_pos_distance (loss/pointnetvlad_loss.py:6-12) given the gradients of (min_pos, max_pos) [bq] each:
 * gq [bq][D], gpos [bq][P][D] (contiguous); q / pos addressed like lpd_metric_loss; P <= 64. */
int lpd_best_pos_bwd(const float* q, long long q_sb, const float* pos, long long pos_sb, long long pos_st, const float* gmin,
                     const float* gmax, int bq, int P, int D, float* gq, float* gpos, void* stream);

/*
 * Descriptor retrieval (evaluate.py:162-206: KDTree(database).query(query, k = 25) per query): the k nearest database
 * descriptors of every query by squared Euclidean distance, ascending, ties -> lower index.
 *   S [nq][ndb] = Q D^T (row-major, from lpd_gemm), Q [nq][ldq], D [ndb][ldd] the descriptors (for the norms),
 *   idx [nq][k] int32, dist [nq][k] squared distances, ws nq + ndb floats.
 */
int lpd_retrieval_topk(const float* S, const float* Q, int ldq, const float* D, int ldd, int nq, int ndb, int dim, int k,
                       int32_t* idx, float* dist, float* ws, void* stream);

/*
 * Batched hard-negative selection (util/data.py:103-115, called for every item of the second training phase with 4000 sampled
 * negatives: a KDTree per query there): for query b, among the rows cand[b][0..nc) of the latent-vector table, the k nearest to
 * Q[b] by squared Euclidean distance, nearest first, as POSITIONS into cand[b] (ties -> lower position); dist [bq][k].
 *   table [n_items][ldt] (the descriptors of the whole training set, kept on the device), Q [bq][ldq], cand [bq][nc] int32.
 */
int lpd_hard_negatives(const float* table, long long ldt, const float* Q, long long ldq, const int32_t* cand, int bq, int nc, int dim,
                       int k, int32_t* pos, float* dist, void* stream);

/* float64 -> float32, n elements (submap files are float64, loading_pointclouds.py:26-35; evaluate.py:115 `.float()`). */
int lpd_f64_to_f32(const double* in, float* out, long long n, void* stream);

/* Per-cloud Morton (Z-order) reordering of the input points: out[b][r] = xyz[b][perm[b][r]].  The descriptor is
 * invariant to point order; sorting makes the neighbour gathers of the aggregation kernels cache-local.
 * xyz/out [B][N][3] (out != xyz), perm [B][N] int32 or NULL.  N <= 16384. */
int lpd_morton_sort(const float* xyz, float* out, int32_t* perm, int B, int N, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path (forward in train mode + backward).  What `loss.backward()` does implicitly in the
 * reference (train_pointnetvlad.py:129,158) through BatchNorm batch statistics, LeakyReLU, the max
 * over k, the neighbour gather, softmax and the NetVLAD normalisations; dense products reuse lpd_gemm.
 * fp64 buffers hold the statistics / reduction results ([C] doubles, caller-owned).
 * ------------------------------------------------------------------------------------------------ */

/* Column sums and sums of squares over R rows of X [R][ld] (BatchNorm batch statistics,
 * nn.BatchNorm1d/2d in train mode).  C and ld multiples of 4. */
int lpd_colstats(const float* X, long long ld, long long R, int C, double* sum, double* sumsq, double* stat_ws, void* stream);

/* From the sums: mean, biased variance -> scale = gamma/sqrt(var+eps), shift = beta - mean*scale, mean, invstd;
 * updates running_mean / running_var in place (momentum, unbiased variance) when given. */
int lpd_bn_finalize(const double* sum, const double* sumsq, double count, int C, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps, float* scale, float* shift,
                    float* mean, float* invstd, void* stream);

/* Y = act(scale * X + shift) elementwise per column (scale/shift NULL = identity affine).  In-place allowed. */
int lpd_affine_act(const float* X, long long ldx, float* Y, long long ldy, long long R, int C, const float* scale,
                   const float* shift, int act, float slope, void* stream);
/* the same with a bf16 copy of the result rows beside the fp32 ones (Y16 [R][ld16] in bf16 elements, ld16 % 4 == 0); Y may be NULL: only
   the bf16 rows are written */
int lpd_affine_act2(const float* X, long long ldx, float* Y, long long ldy, void* Y16, long long ld16, long long R, int C, const float* scale,
                    const float* shift, int act, float slope, void* stream);

/* Backward of Y = act(BN(X)) given dY: dbeta = sum dpre, dgamma = sum dpre*xhat (fp64 [C]) and
 * dX = scale*(dpre - dbeta/R - xhat*dgamma/R); has_bn = 0: plain activation backward (dbeta = bias gradient).
 * X is the raw pre-BatchNorm tensor.  In-place (dX == dY) allowed. */
int lpd_bn_act_bwd(const float* dY, long long lddy, const float* X, long long ldx, float* dX, long long lddx, long long R,
                   int C, const float* scale, const float* shift, const float* mean, const float* invstd, int act,
                   float slope, int has_bn, double* dbeta, double* dgamma, double* stat_ws, void* stream);
/* the same on bf16 tensors (dY, X, dX [R][ld] in bf16 elements, ld % 8 == 0, C a power of two in 8..2048; dX may alias dY): the
 * bf16-storage training mode's conv3 map (util/lpdnet_model.py:262 backward) */
int lpd_bn_act_bwd_bf16(const void* dY, long long lddy, const void* X, long long ldx, void* dX, long long lddx, long long R, int C,
                        const float* scale, const float* shift, const float* mean, const float* invstd, int act, float slope, int has_bn,
                        double* dbeta, double* dgamma, double* stat_ws, void* stream);

/* Materialised edge tensor (training only): U[(i,t)] = P[nbr(i,t)] + Q[i], rows i*k+t, [M*k][C]
 * (the split form of util/lpdnet_model.py:350-357 + the 1x1 conv).  C in {64,128,256}.
 * sum / sumsq ([C] doubles, both or neither): the BatchNorm statistics of U, accumulated while the rows are written
 * (what lpd_colstats would compute in a second pass over U). */
int lpd_edge_build(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx, float* U, long long M,
                   int N, int C, int k, double* sum, double* sumsq, double* stat_ws, void* stream);

/* out[i][c] = act(scale[c] * sel_t X[(i,t)][c] + shift[c]) over k consecutive rows (x.max(dim=-1) after
 * BatchNorm + activation, lpdnet_model.py:250,252,258); arg[i][c] = selected t (uint8). */
int lpd_group_max(const float* X, long long ldx, int k, const float* scale, const float* shift, int act, float slope,
                  float* out, long long ldo, uint8_t* arg, long long M, int C, void* stream);
/* The same, also keeping the RAW selected values xsel[i][c] = X[(i,arg[i][c])][c] ([M][ldsel]) for lpd_edge_bn_bwd_sel. */
int lpd_group_max_sel(const float* X, long long ldx, int k, const float* scale, const float* shift, int act, float slope,
                      float* out, long long ldo, uint8_t* arg, float* xsel, long long ldsel, long long M, int C, void* stream);

/* Backward of lpd_group_max w.r.t. the post-activation edge values: dX[(i,arg[i][c])][c] (+)= dOut[i][c];
 * accumulate = 0 writes all k rows (zeros elsewhere). */
int lpd_group_max_bwd(const float* dOut, long long ldo, const uint8_t* arg, int k, float* dX, long long M, int C,
                      int accumulate, void* stream);

/* Fused backward of out[i] = max_t act(BN(X[(i,t)])) on a materialised edge tensor X [M*k][C]: the arg-max gradient
 * comes from (dOut [M][ldo], arg [M][C]); dDense [M*k][C] (optional) is an additional dense gradient on the
 * post-activation edge values (DG1: the DG2 convolution consumes every edge).  Writes dX [M*k][C] (may alias dDense),
 * dQ[i] = sum_t dX[(i,t)] (optional, the centre-term gradient) and the fp64 reductions dbeta/dgamma [C]. */
int lpd_edge_bn_bwd(const float* dOut, long long ldo, const uint8_t* arg, const float* dDense, const float* X, float* dX,
                    float* dQ, long long ldq, int k, long long M, int C, const float* scale, const float* shift,
                    const float* mean, const float* invstd, int act, float slope, float inv_ns, double* dbeta, double* dgamma,
                    double* stat_ws, void* stream);
/* inv_ns (lpd_edge_bn_bwd, lpd_edge_bn_bwd_bf16): 0 = X holds the raw pre-BatchNorm values.  > 0 (dense form, act none / LeakyReLU with
 * negative slope 1 / inv_ns): X holds the POST-activation values Y = act(BN(U)) -- what lpd_edge_mlp_train stores instead of U -- and
 * the pre-activation is recovered as y (y > 0) or y * inv_ns; `mean` then carries the BatchNorm bias beta and `invstd` 1 / gamma
 * (xhat = (pre - beta) / gamma), `scale` stays gamma * invstd.
 * The arg-max-only form (no dense gradient, no dQ) with the raw selected values Xsel [M][ldsel] the forward kept
 * (lpd_group_max_sel): the dbeta / dgamma reduction is an [M][C] pass instead of a gather from the edge tensor. */
int lpd_edge_bn_bwd_sel(const float* dOut, long long ldo, const uint8_t* arg, const float* X, const float* Xsel, long long ldsel,
                        float* dX, int k, long long M, int C, const float* scale, const float* shift, const float* mean,
                        const float* invstd, int act, float slope, double* dbeta, double* dgamma, double* stat_ws, void* stream);

/* dQ[i] = sum_t dU[(i,t)]  (gradient of the centre term). */
int lpd_group_sum(const float* dU, int k, float* dQ, long long ldq, long long M, int C, void* stream);

/* dP[nbr(i,t)] += dU[(i,t)]  (transpose of the neighbour gather; float atomics; dP zeroed by the caller). */
int lpd_scatter_add_rows(const float* dU, const int32_t* idx, float* dP, long long ldp, long long M, int N, int k, int C,
                         void* stream);

/* Transposed kNN graph in CSR form: rowptr [M+1], edges [M*k] (edge id = i*k + t, grouped by the neighbour row they
 * point to); ws: 2*M int32 scratch.  Built once per graph and training step. */
int lpd_graph_transpose(const int32_t* idx, long long M, int N, int k, int32_t* rowptr, int32_t* edges, int32_t* ws,
                        void* stream);
/* dP[j] (+)= sum of dU[e] over the incoming edges e of row j: the same result as lpd_scatter_add_rows without float
 * atomics (each dU row is read once).  dU [M*k][C] contiguous, C in {64,128,256}. */
int lpd_gather_sum_rows(const float* dU, const int32_t* rowptr, const int32_t* edges, float* dP, long long ldp, long long M,
                        int C, int accumulate, void* stream);

/* dW[o][c] = sum_m dY[m][o] * X[m][c] for Kin <= 8 input channels (first layer weight gradient). */
int lpd_dw_smallk(const float* dY, long long lddy, const float* X, long long ldx, long long M, int Co, int Kin, float* dW,
                  void* stream);

/* Per-cloud max over the N points with the arg-max row (first maximum): in [B][N][ld] -> out [B][C], arg [B][C]
 * (train-mode MaxPool2d((num_points,1)) / torch.max(x, 2): util/PointNetVlad.py:162, util/lpdnet_model.py:300). */
int lpd_colmax_arg(const float* in, long long ld, float* out, int32_t* arg, int B, int N, int C, void* stream);

/* Backward of lpd_colmax_arg: dIn[b*N + arg[b][c]][c] = dOut[b][c]; dIn [B*N][ld] zero-filled by the caller. */
int lpd_colmax_bwd(const float* dOut, const int32_t* arg, float* dIn, long long ld, int B, int N, int C, void* stream);

/* Gradient of a per-cloud KD x KD alignment matrix (KD <= 8): dT[b] = sum over the cloud's points of X[m]^T dY[m]
 * (backward of x @ trans, util/lpdnet_model.py:229, util/PointNetVlad.py:209). */
int lpd_cloud_outer(const float* X, long long ldx, const float* dY, long long ldy, float* dT, int B, int N, int KD, void* stream);

/* Softmax backward with the a_sum path folded in: dS = A*(g - sum_c A*g), g = dA + dasum[cloud]. */
int lpd_softmax_bwd(const float* A, const float* dA, const float* dasum, float* dS, long long rows, int ncols,
                    int rows_per_cloud, void* stream);

/* Backward of lpd_vlad_finalize: dVraw [B][F][KC], dasum [B][KC], dcw2 [F][KC] (summed over clouds). */
int lpd_vlad_finalize_bwd(const float* dOut, const float* v, const float* inv_c, const float* inv_g, const float* asum,
                          const float* cw2, float* dVraw, float* dasum, float* dcw2, int B, int F, int KC, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Training path, second generation (csrc/lpd_train2.hip).
 * ------------------------------------------------------------------------------------------------ */

/*
 * SPLIT-FORM edge stage in train mode: x3 = max_k act(BN_train(convSN1(cat(f_j, f_i)))) (util/lpdnet_model.py:256-258)
 * without the [M*k][C] edge tensor U[(i,t)] = P[nbr(i,t)] + Q[i].  One gather pass:
 *   S[i] = sum_t P[nbr(i,t)];  usel[i] = sel_t P[nbr(i,t)] + Q[i] (sel = max where gamma[c] >= 0, min where < 0 -- the sign of
 *   the BatchNorm scale), arg[i] = the selected slot;  sum / sumsq [C] doubles = the batch statistics of U in closed form
 *   (sum U = sum_i (S_i + k Q_i), sum U^2 = sum_i (sum_t P_nbr^2 + 2 Q_i S_i + k Q_i^2)) for lpd_bn_finalize with count M*k.
 * x3 = act(scale * usel + shift) is then lpd_affine_act on [M][C].  S, usel [M][C] contiguous; C in {64,128,256}; k <= 255.
 */
int lpd_edge_split_fwd(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx, const float* gamma,
                       float* S, float* usel, uint8_t* arg, long long M, int N, int C, int k, double* sum, double* sumsq,
                       double* stat_ws, void* stream);
/*
 * Train-mode DG1 -> DG2 stage in ONE launch (util/lpdnet_model.py:249-252 under batch statistics), replacing
 * lpd_edge_build + lpd_edge_act_max + the [E,128] x [128,128] product + lpd_group_max / lpd_group_sel_stats_bf16:
 *   Y1e[(i,t)][:] = act(s1 (P[nbr(i,t)] + Q[i]) + b1)   (s1 / b1 = BatchNorm1 scale / shift of THIS batch: lpd_edge_split_fwd gives the
 *                   statistics of U = P[nbr] + Q without writing U, and x1 = max_t Y1e with its arg-max through act(s1 usel + b1))
 *   Z = Y1e W2^T (raw), sum / sumsq = its column statistics (fp64, zeroed here), zsel[i][c] = max_t Z (gamma2[c] >= 0) or min_t Z,
 *   arg2[i][c] = the first slot t that attains it.
 * bf16 != 0: Y1e is a bf16 tensor (bf16 storage mode) and the product takes the rounded Y1e (two MFMA products with the split
 * weight); else fp32 and three split-bf16 products.  z_bf16 != 0: Z is stored as bf16 (required with bf16 Y1e, the default with
 * fp32 Y1e too: statistics and selection come from the fp32 accumulators, the stored Z only feeds the backward's xhat2 m2 term).
 * 128 -> 128 channels, M % 64 == 0, N % 64 == 0, k <= 255.
 */
int lpd_edge_mlp_train(const float* P, int ldp, const float* Q, int ldq, const int32_t* idx, const float* s1, const float* b1,
                       const float* W2, const float* gamma2, void* Y1e, void* Z, int bf16, int z_bf16, float* zsel, int ldsel, uint8_t* arg2,
                       double* sum, double* sumsq, int M, int N, int k, int act, float slope, double* stat_ws, void* stream);
/*
 * Backward of the train-mode DG1 -> DG2 stage (the counterpart of lpd_edge_mlp_train), two launches instead of the chain
 * product -> reduce -> apply -> gather over [E,128] tensors:
 *   lpd_edge_mlp_train_bwd: dZ generated from the stored Z while the operand tile is built (A = z a1 + a0 + delta dpre2, cf.
 *     lpd_gemm_bf16s_bnbwd), dY1e = dZ W2 on the MFMA, and in its epilogue G = (dY1e + delta_{t,arg1} dx1) act'(pre1) -- the gradient in
 *     front of BatchNorm1 -- written INSTEAD of dY1e, with dbeta1 = sum G, dgamma1 = sum G xhat1 and gsum[i] = sum_t G[(i,t)];
 *     act' and xhat1 = (pre1 - beta1) rgamma1 come from the stored post-activation Y1e (pre1 = y or y * inv_ns).
 *   lpd_edge_dense_bwd_apply: dP, dQ of U = P[nbr] + Q in closed form from ONE gather pass over the transposed graph (a G row and a Q
 *     row per edge): dP_j = s (sum G - deg m1 - m2 invstd (deg (P_j - mu) + sum Q_i)), dQ_j = s (gsum_j - k m1 - m2 invstd (S_j + k (Q_j - mu))).
 * bf16 != 0: Y1e, dpre2 and G are bf16; z_bf16 != 0: Z is bf16 (as lpd_edge_mlp_train stored it).  M % 32 == 0, 128 channels, k <= 255.
 * Z may be NULL in both calls: lpd_edge_mlp_train then stores no Z, and lpd_edge_mlp_train_bwd forms the dense part of BatchNorm2's
 *     backward as Y1e K (K = W2^T diag(-invstd2 m2 s2) W2, one bf16 product) from `kws` (128 * 128 + 128 floats of caller scratch).
 */
int lpd_edge_mlp_train_bwd(const void* Z, const uint8_t* arg2, const void* dpre2, const float* W2, const float* scale2, const float* mean2,
                           const float* invstd2, const double* dbeta2, const double* dgamma2, const void* Y1e, const uint8_t* arg1,
                           const float* dx1, int lddx1, const float* beta1, const float* rgamma1, int bf16, int z_bf16, void* G, float* gsum,
                           double* dbeta1, double* dgamma1, int M, int k, int act, float slope, float inv_ns, float* kws, double* stat_ws, void* stream);
int lpd_edge_dense_bwd_apply(const void* G, int bf16, const float* gsum, const float* S, const float* P, long long ldp, const float* Q,
                             long long ldq, const int32_t* rowptr, const int32_t* edges, float* dP, long long lddp, float* dQ, long long lddq,
                             long long M, int C, int k, const float* scale, const float* mean, const float* invstd, const double* dbeta,
                             const double* dgamma, void* stream);
/* The same on cloud-resident slices (the organisation of lpd_edge_gather_max16: a block holds an 8-channel slice of a whole cloud in
 * LDS and gathers the k neighbour pieces from there): idx16 from lpd_pack_idx16; k = 20, N <= 4096, C % 8 == 0.  S, usel, arg are
 * bit-identical to lpd_edge_split_fwd, the statistics equal up to the order of the fp64 additions. */
int lpd_edge_split_fwd16_applies(int N, int C, int k);
int lpd_edge_split_fwd16(const float* P, long long ldp, const float* Q, long long ldq, const uint16_t* idx16, const float* gamma, float* S,
                         float* usel, uint8_t* arg, long long M, int N, int C, int k, double* sum, double* sumsq, double* stat_ws, void* stream);

/*
 * Backward of the split-form stage: dOut [M][ldo] = gradient of x3.  half != 0 (bf16 storage, C = 256): the rows the second pass gathers
 * over the transposed graph -- G and Q -- are bf16 copies made by the first pass, kept in the same scratch ([2][M][C] bf16); half & 2:
 * dP / dQ (passed as float*) are bf16 rows as well, lddp / lddq in bf16 elements.
 * G [M][C] (scratch, receives dpre = dOut * act'),
 * dbeta / dgamma [C] doubles (the BatchNorm parameter gradients), dP [M][lddp] and dQ [M][lddq]:
 *   dQ_i = s (dpre_i - k m1 - m2 invstd (S_i + k (Q_i - mu)))
 *   dP_j = s (A_j - deg_j m1 - m2 invstd (deg_j (P_j - mu) + R_j)),  A_j / R_j summed over the incoming edges of j in the
 *   transposed graph (rowptr, edges from lpd_graph_transpose), m1 = dbeta / (M k), m2 = dgamma / (M k).  No float atomics.
 */
int lpd_edge_split_bwd(const float* dOut, long long ldo, const float* usel, const uint8_t* arg, const float* S, const float* P,
                       long long ldp, const float* Q, long long ldq, const int32_t* rowptr, const int32_t* edges, float* G,
                       float* dP, long long lddp, float* dQ, long long lddq, long long M, int C, int k, const float* scale,
                       const float* shift, const float* mean, const float* invstd, int act, float slope, int half, double* dbeta,
                       double* dgamma, double* stat_ws, void* stream);

/*
 * bf16 STORAGE of the per-edge tensors that must exist (BASELINE.json configs[2]: the DG1 -> DG2 chain, where convDG2
 * consumes every post-activation edge, lpdnet_model.py:249-252).  bf16 tensors are uint16_t* ([rows][C] contiguous);
 * statistics, reductions and accumulations stay fp32 / fp64.
 */
/* lpd_edge_build with a bf16 result; the statistics are those of the stored (rounded) values */
int lpd_edge_build_bf16(const float* P, long long ldp, const float* Q, long long ldq, const int32_t* idx, uint16_t* U, long long M,
                        int N, int C, int k, double* sum, double* sumsq, double* stat_ws, void* stream);
/* one pass over U: Y = act(scale U + shift) (bf16, the dense consumer's input) and out[i] = act(scale sel_t U + shift), arg[i] */
/* fp32 storage form of the same pass: Y [M*k][C] = act(scale * U + shift), out [M][ldo] = max over the k rows of a point, arg the slot */
int lpd_edge_act_max(const float* U, int k, const float* scale, const float* shift, int act, float slope, float* Y, float* out,
                     long long ldo, uint8_t* arg, long long M, int C, void* stream);
int lpd_edge_act_max_bf16(const uint16_t* U, int k, const float* scale, const float* shift, int act, float slope, uint16_t* Y,
                          float* out, long long ldo, uint8_t* arg, long long M, int C, void* stream);
/* one pass over the raw conv output Z: batch statistics (sum, sumsq) and the raw selected value sel[i] = sel_t Z[(i,t)] with its
 * slot (max where gamma >= 0, else min); BatchNorm + activation of sel is an [M][C] lpd_affine_act afterwards */
int lpd_group_sel_stats_bf16(const uint16_t* Z, int k, const float* gamma, float* sel, long long lds, uint8_t* arg, long long M, int C,
                             double* sum, double* sumsq, double* stat_ws, void* stream);
/* lpd_edge_bn_bwd on bf16 tensors (dDense optional; dX may alias dDense) */
int lpd_edge_bn_bwd_bf16(const float* dOut, long long ldo, const uint8_t* arg, const uint16_t* dDense, const uint16_t* X, uint16_t* dX,
                         float* dQ, long long ldq, int k, long long M, int C, const float* scale, const float* shift,
                         const float* mean, const float* invstd, int act, float slope, float inv_ns, double* dbeta, double* dgamma,
                         double* stat_ws, void* stream);
/* arg-max-only form with the raw selected values of lpd_group_sel_stats_bf16 (cf. lpd_edge_bn_bwd_sel) */
int lpd_edge_bn_bwd_bf16_sel(const float* dOut, long long ldo, const uint8_t* arg, const uint16_t* X, const float* Xsel, long long ldsel,
                             uint16_t* dX, int k, long long M, int C, const float* scale, const float* shift, const float* mean,
                             const float* invstd, int act, float slope, double* dbeta, double* dgamma, double* stat_ws, void* stream);
/* Backward of x2 = max_k act(BN_train(Z)), Z = Y1e W2^T (lpdnet_model.py:251-252) on bf16 edge tensors WITHOUT the [M*k][128] gradient
 * dZ (csrc/lpd_train3.hip).  (1) dpre16 [M][C] = bf16(dOut * act'(scale Xsel + shift)) and the fp64 sums dbeta = sum dpre,
 * dgamma = sum dpre xhat, from the raw selected values Xsel [M][ldsel] of lpd_group_sel_stats_bf16. */
int lpd_bn_sel_bwd_reduce(const float* dOut, long long ldo, const float* Xsel, long long ldsel, long long M, int C, const float* scale,
                          const float* shift, const float* mean, const float* invstd, int act, float slope, uint16_t* dpre16,
                          double* dbeta, double* dgamma, double* stat_ws, void* stream);
/* (2) dW2 [128][128] = dZ^T Y1e from ONE pass over Y1e [M*k][128] bf16: S = D^T Y1e (D: dpre16 at the arg-max slots), the Gram matrix
 * Y1e^T Y1e and the column sums, then dW2[c] = s_c (S[c] - m1_c s - m2_c invstd_c (W2[c] G - mu_c s)) in fp64.  W2 [128][ldw]: the
 * convolution weight; dbeta / dgamma: the sums of (1); ws: lpd_edge_dw_sel_bf16_ws_bytes(M * k) bytes, 16-byte aligned.
 * Requires (M * k) % 128 == 0, 8 <= k <= 255. */
long long lpd_edge_dw_sel_bf16_ws_bytes(long long E);
int lpd_edge_dw_sel_bf16(const uint16_t* Y, const uint8_t* arg, const uint16_t* dpre16, int k, long long M, const float* W2, long long ldw,
                         const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma, float* dW2,
                         void* ws, void* stream);
/* (3) dY [M*k][128] (bf16) = dZ W2 with dZ = s (delta dpre - m1 - xhat m2) generated from Z [M*k][128] (bf16), arg [M][128] and dpre16
 * while the operand is loaded.  16 <= k <= 255. */
int lpd_gemm_bf16s_bnbwd(const uint16_t* Z, const uint8_t* arg, const uint16_t* dpre16, int k, long long M, const float* W2, int ldw,
                         const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma, uint16_t* dY,
                         void* stream);
/* The three steps above for fp32 STORAGE (fp32 Y1e / Z / dY, fp32 dpre; split-bf16 products): (M * k) % 64 == 0, 16 <= k <= 255; the
 * workspace of lpd_edge_dw_sel_f32 is lpd_edge_dw_sel_bf16_ws_bytes(M * k) bytes. */
int lpd_bn_sel_bwd_reduce_f32(const float* dOut, long long ldo, const float* Xsel, long long ldsel, long long M, int C, const float* scale,
                              const float* shift, const float* mean, const float* invstd, int act, float slope, float* dpre, double* dbeta,
                              double* dgamma, double* stat_ws, void* stream);
int lpd_edge_dw_sel_f32(const float* Y, const uint8_t* arg, const float* dpre, int k, long long M, const float* W2, long long ldw,
                        const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma, float* dW2,
                        void* ws, void* stream);
int lpd_gemm_f32s_bnbwd(const float* Z, const uint8_t* arg, const float* dpre, int k, long long M, const float* W2, int ldw,
                        const float* scale, const float* mean, const float* invstd, const double* dbeta, const double* dgamma, float* dY,
                        void* stream);
/* lpd_gather_sum_rows with a bf16 edge-gradient tensor (fp32 sums) */
int lpd_gather_sum_rows_bf16(const uint16_t* dU, const int32_t* rowptr, const int32_t* edges, float* dP, long long ldp, long long M,
                             int C, int accumulate, void* stream);
/* C [M][N] (bf16) = A [M][K] (bf16) x W, W fp32 [N][ldw] (b_kmajor = 0, torch conv weight) or [K][ldw] (1): the weight is split
 * hi + lo (two products on v_mfma_f32_32x32x16_bf16, fp32 accumulation).  (N, K) in {(128,128), (64,64)}. */
int lpd_gemm_bf16s(const uint16_t* A, const float* W, int ldw, int b_kmajor, uint16_t* C, long long M, int N, int K, void* stream);
/* dW [KA][KB] (fp32) = sum_m A[m][:]^T B[m][:], A [M][KA], B [M][KB] bf16; ws: lpd_gemm_tn_bf16_ws_floats(M, KA, KB) floats */
long long lpd_gemm_tn_bf16_ws_floats(long long M, int KA, int KB);
int lpd_gemm_tn_bf16(const uint16_t* A, const uint16_t* B, float* dW, float* ws, long long M, int KA, int KB, void* stream);

/* dW [KA][KB] (fp32) = sum_m A[m][:]^T B[m][:] for fp32 operands A [M][lda], B [M][ldb] in split-bf16 form (three MFMA products per
 * term, fp32-grade): the weight gradients dW = dY^T X of the training path and, batched over the clouds, the NetVLAD residual
 * pooling act^T x (util/PointNetVlad.py:64-67).  batch problems at strides sA / sB (elements) write dW [batch][KA][KB].
 * KA %% 128 == 0, KB %% 64 == 0; ws: lpd_gemm_tn_ws_floats(M, KA, KB, batch) floats.
 * bf16_rows is a flag word: bit 0 = A holds bf16 rows (lda / sA in bf16 elements; they are the hi image: two products), bit 1 = B holds
 * bf16 rows as well (ldb in bf16 elements; needs bit 0, KA % 256 == 0, KB % 256 == 0, M % 32 == 0, M >= 2048, batch 1): one product per
 * term, exact in the operands.  Products with KA % 256 == 0 over whole 32-row chunks run on row-major LDS images read with
 * ds_read_b64_tr_b16, the others on the register-transposing kernels.
 */
long long lpd_gemm_tn_ws_floats(long long M, int KA, int KB, int batch);
int lpd_gemm_tn(const void* A, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB,
                int batch, long long sA, long long sB, int bf16_rows, void* stream);
/* lpd_gemm_tn with the rows of A taken as act(a_scale[a] A[m][a] + a_shift[a]) (multiply, then add: lpd_affine_act's bits; bf16 rows:
 * rounded to bf16 again): a train-mode BatchNorm affine + activation applied where the raw map is staged, so that the activated
 * [B N, 1024] map of util/lpdnet_model.py:262 need not be stored for util/PointNetVlad.py:64-67 (pooling) and the assignment's weight
 * gradient.  KA % 256 == 0, KB % 64 == 0 and KB % 128 != 0, M % 32 == 0, M >= 2048; workspace: lpd_gemm_tn_act_ws_floats. */
long long lpd_gemm_tn_act_ws_floats(long long M, int KA, int KB, int batch, int a_bf16);
int lpd_gemm_tn_act(const void* A, long long lda, const float* B, long long ldb, float* dW, float* ws, long long M, int KA, int KB, int batch,
                    long long sA, long long sB, int a_bf16, const float* a_scale, const float* a_shift, int a_act, float a_slope, void* stream);
/* a_bf16 != 0: A holds bf16 rows (lda, sA in bf16 elements, multiples of 8): two MFMA products per term (bf16-storage training mode). */

/*
 * conv3_lpd of the eval path (util/lpdnet_model.py:262: 512 -> emb_dims per point, + bn3 + activation) on PRE-SPLIT operands:
 *   C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]),   A = a_hi + a_lo,  three bf16 MFMA products per term (fp32-grade); an
 *   eval-mode BatchNorm rides as W = diag(scale) W_conv, bias = shift
 * a_hi / a_lo: bf16 cloud panels [cloud][K/8][a_panel_ld][8] (hi = bf16(x), lo = bf16(x - hi); written by the producers of
 * [x1 | x2 | x3] or by lpd_split_panels), a_cloud bf16 elements apart; frags: lpd_gemm_prep_b(W [N][K], b_kmajor = 0);
 * C: row-major [M][ldc] (c_cloud = 0) or fp32 cloud panels [cloud][N/8][c_panel_ld][8], c_cloud floats apart.
 * 256 x 256 block tiles, LDS-DMA ring, persistent workgroups (csrc/lpd_gemm_p8.hip).  lpd_gemm_p8_applies: N % 256 == 0,
 * K % 32 == 0, K >= 64, clouds of a multiple of 256 points.  act none / ReLU / LeakyReLU (0 <= slope <= 1).  impl: 0 (= 6) or
 * 5 / 6 staging units in flight.
 */
int lpd_gemm_p8_applies(int M, int N, int K, int panel_n);
int lpd_gemm_p8(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* bias,
                int act, float slope, int impl, void* stream);
/* lpd_gemm_p8 + the NetVLAD assignment product (util/PointNetVlad.py:48, x . cluster_weights) of its own output in the same launch:
 * besides C, parts[j][m][0..64) = C[m][256 j .. 256 j + 255] . W2[256 j .. 256 j + 255][0..64) for the N / 256 column blocks j
 * (part_stride floats between the planes); lpd_softmax_affine_parts sums the planes.  w2_frags: lpd_gemm_prep_b(W2 [N][64], ldb,
 * b_kmajor = 1, 64, N). */
int lpd_gemm_p8_fused(const void* a_hi, const void* a_lo, long long a_cloud, int a_panel_ld, const void* frags, float* C, int ldc,
                      long long c_cloud, int c_panel_ld, int M, int N, int K, int panel_n, const float* bias, int act, float slope,
                      const void* w2_frags, float* parts, long long part_stride, void* stream);
/* softmax(scale * (sum of `parts` planes of in, part_stride floats apart) + shift) over 64 columns, with the per-group column sums
 * of lpd_softmax_affine (group_rows % 64 == 0); parts in {1, 2, 4, 8} */
int lpd_softmax_affine_parts(const float* in, int parts, long long part_stride, float* out, int rows, const float* scale,
                             const float* shift, int group_rows, float* colsum, int colsum_ld, void* stream);
/* fp32 cloud panels [clouds][panels][s_panel_ld][8] (s_cloud floats apart) -> the two bf16 planes hi / lo of the same layout
 * ([clouds][panels][d_panel_ld][8], d_cloud elements apart); n rows per panel are converted. */
int lpd_split_panels(const float* src, long long s_cloud, int s_panel_ld, void* hi, void* lo, long long d_cloud, int d_panel_ld,
                     int clouds, int panels, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LPD_HIP_H */
