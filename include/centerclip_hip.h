/*
 * centerclip_hip.h - C ABI of libcenterclip_hip.so (gfx950 / MI355X).
 *
 * The reference (mzhaoshuai/CenterCLIP) is pure Python/PyTorch: it has no FFI or
 * operator registry, its boundary is a set of Python call signatures (SURVEY.md
 * §8b).  This header declares the C entry points a maintainer of the reference
 * binds (ctypes stub in INTEGRATION.md) to replace those call sites.  Every
 * declaration cites the reference interface it replaces (paths relative to the
 * reference repository root).
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers unless the
 *     name ends in _host; `stream` is a hipStream_t passed as void* (NULL = the
 *     null stream); nothing in here allocates, synchronises the device or
 *     throws: workspace is caller-owned (size it with the *_workspace_bytes
 *     query), every call only enqueues kernels on `stream`;
 *   - return value: CC_OK or a negative cc_status; cc_status_string() names it;
 *   - fp32 tensors are IEEE binary32, index outputs are int64 (torch.long), as
 *     the reference returns them;
 *   - re-entrant: no global state besides per-kernel function attributes.
 */
#ifndef CENTERCLIP_HIP_H
#define CENTERCLIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cc_status {
    CC_OK = 0,
    CC_ERR_INVALID = -1,      /* bad argument (NULL pointer, K > N, non-positive size ...)      */
    CC_ERR_UNSUPPORTED = -2,  /* legal in the reference but not built here      */
    CC_ERR_WORKSPACE = -3,    /* workspace too small / NULL                                      */
    CC_ERR_HIP = -4           /* a HIP runtime call failed (hipGetLastError after a launch ...)  */
} cc_status;

/* Largest `threshold` for which the k-medoids entry points run the whole selection in ONE launch (reference default 1e-5,
 * fast_kmeans.py:14; the shipped scripts pass 1e-6): every problem iterates to its fixed point, which is the final state of
 * the reference's chunk-mean stop test (fast_kmeans.py:85-88) for any threshold below the distance between two distinct
 * tokens.  A looser threshold ends the reference's loop earlier, at a step that depends on all problems of the split chunk:
 * the entry points then run the literal loop (one iteration of every problem per launch + the chunk's center_shift in
 * ATen's summation order, 2 * iter_limit + 2 launches, split_size <= 1024); a NaN threshold is CC_ERR_INVALID. */
#define CC_KMEDOIDS_MAX_THRESHOLD 1e-5f

/* metric: reference strings 'euclidean' / 'cosine' (modules/cluster/cluster_utils.py:21-33) */
#define CC_METRIC_EUCLIDEAN 0
#define CC_METRIC_COSINE 1

const char* cc_version(void);
const char* cc_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * Token addressing shared by the cluster entry points.
 *
 * A clustering problem p = s*B + b (segment-major, modules/cluster/cluster.py:247-250) has
 * N = fd*n tokens; token j = f*n + i (frame-in-segment major).  Its W floats start at
 *     x + b*stride_b + s*stride_s + f*stride_f + i*stride_i          (strides in floats)
 * which covers
 *   - a contiguous [P,N,W] batch (the argument of batch_fast_kmedoids_with_split):
 *         B=P, S=1, fd=1, n=N, stride_b=N*W, stride_i=W;
 *   - the [L=1+n, B*T, W] (LND) activation TokenClusterInter.forward receives, patch rows
 *     only: x += B*T*W, stride_b=T*W, stride_s=fd*W, stride_f=W, stride_i=B*T*W;
 *   - the same activation stored frame-major ([B*T, L, W], NLD): x += W,
 *     stride_b=T*L*W, stride_s=fd*L*W, stride_f=L*W, stride_i=W.
 * ------------------------------------------------------------------------------------------ */
typedef struct cc_token_layout {
    int32_t B;          /* clips (problems per segment)                      */
    int32_t S;          /* segments per clip (T_new); P = S*B                */
    int32_t fd;         /* frames per segment                                */
    int32_t n;          /* tokens per frame; N = fd*n                        */
    int64_t stride_b, stride_s, stride_f, stride_i;
} cc_token_layout;

/* Bytes of scratch the cluster entry points need for P problems of N tokens. */
size_t cc_cluster_workspace_bytes(int32_t P, int32_t N, int32_t W, int32_t pre_norm);

/* L2 norm of every token, norms [P,N] - replaces torch.norm(X, dim=-1) in KKZ_init
 * (modules/cluster/cluster_utils.py:93). */
int cc_token_norms_f32(const float* x, const cc_token_layout* lay, int32_t W, float* norms,
                       void* ws, size_t ws_bytes, void* stream);

/*
 * C3 - replaces pairwise_distance(data, data, metric, self_nearest, all_negative, p)
 *      modules/cluster/cluster_utils.py:8-43 for the self-distance case the hot path uses
 *      (fast_kmeans.py:61-62).  dist [P,N,N] fp32.  `chunk` = number of consecutive
 *      problems that share one max in the all_negative shift (the reference takes the max
 *      of the whole tensor it is handed = one split chunk; pass P for a single call).
 *      norms_out (optional) [P,N] = L2 norm of every token (cluster_utils.py:93).
 */
/* The same function for two DIFFERENT token sets: x1 [P,N1,W], x2 [P,N2,W] contiguous -> dist [P,N1,N2]; the all_negative
 * shift uses the maximum of the whole tensor (cluster_utils.py:35-36), self_nearest lowers dist[.., j, j] for j < N2 (needs
 * N2 <= N1, as the reference's indexing does).  Not on the retrieval path (the k-medoids only ever passes one set).
 * ws: at least 4 bytes when all_negative. */
int cc_pairwise_distance_cross_f32(const float* x1, const float* x2, int32_t P, int32_t N1, int32_t N2, int32_t W,
                                   int32_t metric, float p, int32_t all_negative, int32_t self_nearest, float* dist,
                                   void* ws, size_t ws_bytes, void* stream);
int cc_pairwise_distance_f32(const float* x, const cc_token_layout* lay, int32_t W,
                             int32_t metric, float p, int32_t all_negative, int32_t self_nearest,
                             int32_t chunk, float* dist, float* norms_out,
                             void* ws, size_t ws_bytes, void* stream);

/*
 * C4+C5 from a finished distance tensor (parity level P0): KKZ init
 * (cluster_utils.py:93,106-118), assignment/update iterations, ascending sort and final
 * re-assignment (fast_kmeans.py:65-97).  dist [P,N,N] is used exactly as given (no shift);
 * norms [P,N] selects the first KKZ medoid (first argmax).  Each problem iterates until
 * its medoid vector is unchanged or iter_limit is reached (equivalence 4 in SURVEY §8a).
 * Outputs: medoids [P,K] int64 (ascending if id_sort), assign [P,N] int64 (may be NULL),
 * iters [P] int32 (may be NULL).
 */
int cc_kmedoids_from_dist_f32(const float* dist, const float* norms, int32_t P, int32_t N, int32_t K,
                              int32_t iter_limit, int32_t id_sort,
                              int64_t* medoids, int64_t* assign, int32_t* iters,
                              void* ws, size_t ws_bytes, void* stream);

/*
 * C2..C5 - replaces batch_fast_kmedoids_with_split(X, K, distance, threshold, iter_limit,
 *      id_sort, norm_p, split_size, pre_norm)  modules/cluster/fast_kmeans.py:14-40 and
 *      batch_fast_kmedoids (:45-97; pass split_size >= P).  The stop test: see CC_KMEDOIDS_MAX_THRESHOLD above (fixed point in
 *      one launch up to it, the reference's literal chunk-mean test above it).
 */
int cc_batch_kmedoids_f32(const float* x, const cc_token_layout* lay, int32_t W, int32_t K,
                          int32_t metric, float norm_p, float threshold, int32_t iter_limit,
                          int32_t id_sort, int32_t split_size, int32_t pre_norm,
                          int64_t* medoids, int64_t* assign, int32_t* iters,
                          void* ws, size_t ws_bytes, void* stream);

/*
 * C1..C6 - replaces TokenClusterInter.forward (kmediods++ branch, aggregation=None)
 *      modules/cluster/cluster.py:206-216,239-260,287-289,303-310,350-352.
 * Input: activation with 1+n tokens per frame, B*T frames, W floats per token; token l of
 * frame c starts at x + l*in_tok_stride + c*in_frame_stride (LND: tok=B*T*W, frame=W;
 * frame-major: tok=W, frame=(1+n)*W).  Output: 1+K tokens per segment, B*T_new segments,
 * same addressing with out_*_stride; out token 0 = mean of the segment's fd CLS tokens
 * (cluster.py:307-308), out token 1+k = the k-th medoid token (ascending ids).
 * medoids [T_new*B, K] int64 in the reference's problem order p = s*B + b; assign / iters
 * optional as above.
 */
int cc_token_cluster_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride,
                         int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                         int32_t metric, float norm_p, float threshold, int32_t iter_limit,
                         int32_t split_size, int32_t pre_norm,
                         float* out, int64_t out_tok_stride, int64_t out_frame_stride,
                         int64_t* medoids, int64_t* assign, int32_t* iters,
                         void* ws, size_t ws_bytes, void* stream);

/* The other TokenClusterInter variants that share this data layout, modules/cluster/cluster.py:
 *   algorithm CC_CLUSTER_KMEDOIDS + aggregation CC_AGGREGATE_MEDOID  the shipped path (:289), = cc_token_cluster_f32
 *   algorithm CC_CLUSTER_KMEDOIDS + aggregation CC_AGGREGATE_MEAN    output token k = mean of the tokens assigned to
 *                                medoid k (:291-301; an empty cluster yields NaN, as 0/0 does in the reference)
 *   algorithm CC_CLUSTER_POOLING  every token, CLS included, = mean over the segment's frames (:319-324);
 *                                K and the k-medoids arguments are ignored, out is [1+n, B*T_new, W]
 *   cluster_embed  [K, W] fp32 or NULL: learnt embedding added to output tokens 1..K (:304-305)
 *   cls_multiplier [T] fp32 or NULL: per-frame scale of the CLS token before the segment mean (adaptive_cls, :244-245)
 * Sums follow the association of the reference's torch.sum / mean over a non-innermost dimension (running sum
 * folded into a second level every 16 rows), so the outputs are bit-identical to the reference's CPU path.
 *   algorithm CC_CLUSTER_SPARSE_SAMPLING  eval-mode 'sparse_sampling' (:326-343): fixed_ids [K] int64 (device) are the
 *                                ids of token_sparse_sampling(K, fd*n, random_shift=False) (cluster_utils.py:136-170),
 *                                the same for every segment; gather + CLS mean as for medoids */
#define CC_CLUSTER_KMEDOIDS 0
#define CC_CLUSTER_POOLING  1
#define CC_CLUSTER_SPECTRAL 3          /* spectral clustering picks the medoids / the assignment (cluster_algo 'spectral') */
#define CC_CLUSTER_SPARSE_SAMPLING 2   /* fixed_ids [K]: token_sparse_sampling(K, fd*n, random_shift=False) */
#define CC_AGGREGATE_MEDOID 0
#define CC_AGGREGATE_MEAN   1
typedef struct cc_cluster_variant {
    int32_t algorithm;
    int32_t aggregation;
    const float* cluster_embed;
    const float* cls_multiplier;
    const int64_t* fixed_ids;
    /* CC_CLUSTER_SPECTRAL (cluster.py:262-272 -> spectral.py:17-75): the selection is made by spectral clustering of the
     * segment's tokens (graph mode CC_GRAPH_*, sigma, knn_k, optional [N,N] uint8 spatial-temporal mask, sign correction),
     * the k-medoids tail runs on the row-normalised eigenvectors with the call's metric / norm_p / threshold / iter_limit /
     * split_size; gather / cluster means / CLS mean as for CC_CLUSTER_KMEDOIDS.  Needs cc_spectral_workspace_bytes more
     * scratch behind cc_cluster_workspace_bytes. */
    float spectral_sigma;
    int32_t spectral_graph_mode;
    int32_t spectral_knn_k;
    int32_t spectral_correct_sign;
    const uint8_t* spectral_graph;
    /* mean_residual (cluster.py:228-235, clip.py:239-242) - read by the fused encoders only: the block's residual stream
     * restarts from the frame means of every token (the 'pooling' reduction of the block's input) while ln_1 / the attention
     * read the clustered tokens.  Needs an unchanged token count (cluster_tokens[i] == incoming tokens, the reference's
     * assert).  The module-level entry points return the clustered tokens only; their caller forms the means with
     * algorithm = CC_CLUSTER_POOLING on the same input. */
    int32_t mean_residual;
} cc_cluster_variant;
/* The aggregation step of CC_AGGREGATE_MEAN alone, from a given assignment [T_new*B, fd*n] int64 (values 0..K-1;
 * problem p = s*B + b as everywhere) - the counterpart of cc_token_gather_f32 for cluster means
 * (cluster.py:291-310).  variant may be NULL; only its cluster_embed / cls_multiplier are used. */
int cc_token_aggregate_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride,
                           int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                           const int64_t* assign, const cc_cluster_variant* variant,
                           float* out, int64_t out_tok_stride, int64_t out_frame_stride, void* stream);

/* The same step for a selection made elsewhere (cluster_algo 'spectral', cluster.py:262-272 + :287-310): medoids
 * [T_new*B, K] (aggregation None) or assign [T_new*B, fd*n] (cluster means), + the variant's cluster_embed / cls_multiplier. */
int cc_token_apply_selection_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride,
                                 int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                                 const cc_cluster_variant* variant, const int64_t* medoids, const int64_t* assign,
                                 float* out, int64_t out_tok_stride, int64_t out_frame_stride, void* stream);

int cc_token_cluster_variant_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride,
                                 int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                                 int32_t metric, float norm_p, float threshold, int32_t iter_limit,
                                 int32_t split_size, int32_t pre_norm, const cc_cluster_variant* variant,
                                 float* out, int64_t out_tok_stride, int64_t out_frame_stride,
                                 int64_t* medoids, int64_t* assign, int32_t* iters,
                                 void* ws, size_t ws_bytes, void* stream);

/*
 * N4 (training support): gradient of TokenClusterInter.forward (modules/cluster/cluster.py:239-310,319-343) with respect to
 * its input and parameters for a FIXED selection - the reference selects under no_grad (fast_kmeans.py:13,44), so the
 * medoid ids / assignment are constants of the backward pass, exactly what torch.autograd does with the reference module:
 *   medoid gather (:289)            grad_x[token medoid_k] = grad_out[1 + k]
 *   cluster means (:291-301)        grad_x[j] = grad_out[1 + assign_j] / |cluster(assign_j)|     (assign [T_new*B, fd*n])
 *   cluster_embed (:304-305)        grad_cluster_embed[k] = sum over segments of grad_out[1 + k]            ([K, W] or NULL)
 *   CLS mean (* cls_multiplier, :244-245,307-308)   grad_x[cls, frame t] = grad_out[0] / fd (* m_t);
 *                                   grad_cls_mult[t] = sum_{b,w} grad_out[0] / fd * x[cls, frame t]         ([T] or NULL; needs x)
 *   pooling (:319-324)              grad_x[token] = grad_out[token] / fd
 *   sparse_sampling (:326-343)      gather with variant->fixed_ids
 * Every row of grad_x ([1+n, B*T, W] through the strides) is written exactly once, zeros included: no memset, no atomics.
 * medoids [T_new*B, K] as returned by cc_token_cluster_variant_f32 (ignored for mean / pooling / sparse_sampling).
 */
int cc_token_cluster_backward_f32(const float* grad_out, int64_t go_tok_stride, int64_t go_frame_stride,
                                  int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                                  const cc_cluster_variant* variant, const int64_t* medoids, const int64_t* assign,
                                  const float* x, int64_t x_tok_stride, int64_t x_frame_stride,
                                  float* grad_x, int64_t gx_tok_stride, int64_t gx_frame_stride,
                                  float* grad_cluster_embed, float* grad_cls_mult, void* stream);


/*
 * N4: cluster_algo 'spectral' (modules/cluster/spectral.py:17-165), step by step; the whole selection also runs inside
 * cc_token_cluster_variant_f32 / the fused encoders through cc_cluster_variant.algorithm = CC_CLUSTER_SPECTRAL:
 *   cc_spectral_laplacian_f32  W = exp(-|x_i - x_j|^2 / (2 sigma^2)) (constructW 'HeatKernel', spectral.py:79-88, squared
 *                              distances as batched_cdist_l2, cluster_utils.py:121-133), optionally * graph [N,N] uint8
 *                              (spatial_temporal_graph, :139-165); D = diag(W 1); laplacian [P,N,N] = D^-1/2 (D - W) D^-1/2
 *                              (:44-52).  affinity_out [P,N,N] / degree_out [P,N] optional.  ws: cc_cluster_workspace_bytes.
 *   cc_svd_sign_flip_f32       batch_sign_flip_rasmus_bro (:110-137), in place on U [P,M,K] given S [P,K], VT [P,K,N].
 */
int cc_spectral_laplacian_f32(const float* x, const cc_token_layout* lay, int32_t W, float sigma,
                              const uint8_t* graph, float* laplacian, float* affinity_out, float* degree_out,
                              void* ws, size_t ws_bytes, void* stream);
/* constructW graph modes (spectral.py:86-100): heat kernel, or the heat kernel kept where j is among the knn_k largest entries
 * of row i or i among those of row j (mutual: and).  The spatial-temporal mask is applied after either (:104-105). */
#define CC_GRAPH_HEAT_KERNEL 0
#define CC_GRAPH_KNN         1
int cc_spectral_graph_laplacian_f32(const float* x, const cc_token_layout* lay, int32_t W, float sigma, int32_t mode,
                                    int32_t knn_k, int32_t mutual, const uint8_t* graph, float* laplacian,
                                    float* affinity_out, float* degree_out, void* ws, size_t ws_bytes, void* stream);
/*
 * The decomposition of batch_spectral_clustering (spectral.py:54-61): Q [P, N, ldq] (columns 0..K-1, the rest zeroed) = the
 * K eigenvectors of the symmetric positive semi-definite laplacian [P,N,N] with the smallest eigenvalues, in the reference's
 * column order (U[:, :, -K:] of torch.linalg.svd: eigenvalue descending), eigenvalues [P,K] optional, sweeps_out [P]
 * optional (Jacobi sweeps; 0 from the direct solver).  One workgroup per problem.  Direct solver (eig.hip) - Householder
 * tridiagonalisation (fp32), Sturm multi-section and inverse iteration (fp64), back-transformation: N <= 196 with the matrix
 * and then the packed reflectors + K vectors in LDS (K <= 49 at N = 196, K <= 64), 196 < N <= 832 with K <= 192 from a global
 * scratch in ws (ViT-B/16 ships N = 784, K = 160, scripts/activitynet.sh:104-122).  Remaining shapes up to N = 640: batched
 * one-sided Jacobi on 2I - L.
 * correct_sign: batch_sign_flip_rasmus_bro (:110-137) applied (for a symmetric matrix it depends on the vector alone).
 * Parity: eigenpairs to fp32 working precision, the eigenvalues equal the reference's singular values to 1e-5; the vectors
 * equal the reference's up to sign and, where eigenvalues coincide to rounding, up to a rotation inside that eigenspace -
 * which no solver pins.  Row distances of Q, which is all the k-medoids tail uses, are invariant to both
 * as long as the K-th and (K+1)-th eigenvalue are separated.  ws: cc_spectral_embedding_workspace_bytes.
 */
size_t cc_spectral_embedding_workspace_bytes(int32_t P, int32_t N);
/* Extra scratch of a CC_CLUSTER_SPECTRAL token-cluster call, appended to cc_cluster_workspace_bytes(P, N, W, pre_norm). */
size_t cc_spectral_workspace_bytes(int32_t P, int32_t N, int32_t K);
int cc_spectral_embedding_f32(const float* laplacian, int32_t P, int32_t N, int32_t K, int32_t correct_sign,
                              float* Q, int32_t ldq, float* eigenvalues, int32_t* sweeps_out, void* ws, size_t ws_bytes,
                              void* stream);
/* The same with the solver named: CC_EIG_AUTO = the direct solver wherever it applies (cc_spectral_embedding_f32),
 * CC_EIG_JACOBI = the one-sided Jacobi kernel for every shape (what shapes outside the direct solver's scope run; the
 * tests hold both against float64). */
#define CC_EIG_AUTO   0
#define CC_EIG_JACOBI 1
int cc_spectral_embedding_solver_f32(const float* laplacian, int32_t P, int32_t N, int32_t K, int32_t correct_sign,
                                     float* Q, int32_t ldq, float* eigenvalues, int32_t* sweeps_out, int32_t solver,
                                     void* ws, size_t ws_bytes, void* stream);
int cc_svd_sign_flip_f32(float* U, const float* S, const float* VT, int32_t P, int32_t M, int32_t K, int32_t N,
                         void* stream);

/*
 * C6 alone - the gather half of TokenClusterInter.forward for given medoid ids:
 *      x_tmp = res_tmp[batch_index, mediods_ids] (modules/cluster/cluster.py:289), the per-segment
 *      CLS mean (:307-308) and the restack (:303,310).  medoids [T_new*B, K] int64, p = s*B + b.
 */
int cc_token_gather_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride,
                        int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                        const int64_t* medoids, float* out, int64_t out_tok_stride,
                        int64_t out_frame_stride, void* stream);

/* ==========================================================================================
 * CLIP forward path.  fp16 MFMA operands, fp32 accumulate,
 * fp32 residual stream / LayerNorm / softmax (modules/clip.py:183-189 keeps LN in fp32; the
 * reference runs fp16 weights on GPU via convert_weights, clip.py:515-536).
 * Weight tensors keep the reference's state-dict layouts: Linear / in_proj
 * weights [out, in] row-major, conv1 weight [W, 3, p, p] == [W, 3*p*p]; "f16" pointers are IEEE
 * binary16 copies of those tensors, everything else fp32.
 * ========================================================================================== */

/* epilogue ids of cc_linear_f16 */
#define CC_EPI_F16 0        /* C(fp16) = A W^T + b                                           */
#define CC_EPI_F16_GELU 1   /* C(fp16) = QuickGELU(A W^T + b)         modules/clip.py:192-194 */
#define CC_EPI_F32_RESID 2  /* C(fp32) += A W^T + b   (residual add)  modules/clip.py:240,251 */
#define CC_EPI_F32 4        /* C(fp32) = A W^T + b                                           */

/* nn.Linear forward, y = x W^T + b  (c_fc / c_proj / out_proj: modules/clip.py:207-211; the
 * packed in_proj of nn.MultiheadAttention: clip.py:205).  a [M,K] fp16, w [N,K] fp16, bias
 * [N] fp32 or NULL, c row stride ldc elements.  K % 64 == 0, N % 64 == 0.  tile: 0 = auto. */
int cc_linear_f16(const void* a_f16, const void* w_f16, const float* bias, void* c,
                  int32_t M, int32_t N, int32_t K, int32_t ldc, int32_t epilogue, int32_t tile,
                  void* stream);
/* out [M, N] fp32 = resid [M, N] + a w^T + bias (the "f32_resid" epilogue with a separate source of the rows that are added;
 * resid == out is cc_linear_f16's in-place form). */
int cc_linear_resid_f16(const void* a_f16, const void* w_f16, const float* bias, const float* resid, float* out, int32_t M,
                        int32_t N, int32_t K, int32_t tile, void* stream);
/* LayerNorm over the last dim (fp32 statistics, eps as given) - modules/clip.py:183-189.
 * Row r is read at in + r*in_stride and written at out + r*out_stride (elements); out is fp16
 * when out_f16 != 0, else fp32 (may alias in). */
int cc_layernorm_f32(const float* in, int64_t in_stride, const float* gamma, const float* beta,
                     void* out, int64_t out_stride, int32_t rows, int32_t W, float eps,
                     int32_t out_f16, void* stream);

/* The three pieces of the folded-LayerNorm pipeline the encoders use instead of stand-alone LayerNorm passes
 * (replaces ln_1 -> in_proj and ln_2 -> c_fc of modules/clip.py:240,251):
 *   cc_row_stats_f16          h [rows,W] fp32 -> h16 = fp16(h - c), stats [rows][1][2] = (sum, sum of squares) of h16,
 *                             c = the row mean, written to shift_out [rows] (shift_out NULL: c = 0).
 *   cc_linear_ln_f16          out(fp16) = [QuickGELU](LN(h) W^T + b), from h16, the folded weight and the stats
 *                             (LayerNorm is invariant to the per-row shift c, so the consumer never sees it)
 *   cc_linear_resid_stats_f16 h += a W^T + b (residual), h16 = fp16(h - c), stats [M][*slots_out][2] (one slot per
 *                             tile column x wave column; stats buffers must hold 32 slots per row).  c = the row mean
 *                             before this update = shift_in[m] + mean of the centred copy the previous stage wrote
 *                             (stats_in [M][slots_in][2]); written to shift_out [M].  stats_in NULL: c = 0.
 * Centring keeps the rounding error of the fp16 copy relative to the row's spread, not to its mean (rows of real
 * checkpoints can have |mean| >> sigma). */
int cc_row_stats_f16(const float* h, void* h16_out, float* stats_out, float* shift_out, int32_t rows, int32_t W,
                     void* stream);
int cc_linear_ln_f16(const void* h_f16, const void* w_ln_f16, const float* c1, const float* c2,
                     const float* stats, int32_t slots, float eps, void* out_f16,
                     int32_t M, int32_t N, int32_t K, int32_t gelu, int32_t tile, void* stream);
/* in_proj with the LayerNorm folded + the multi-head attention core in ONE launch (modules/clip.py:210-214, the ln_1 ->
 * nn.MultiheadAttention path of ResidualAttentionBlock.attention): att [M, W] fp16 = softmax(q k^T / 8 [causal]) v per
 * (sequence, head) with q | k | v = LN(h) Wqkv^T + b evaluated as in cc_linear_ln_f16 (w_ln_f16 [3W, W], c1 / c2 [3W], stats
 * [M][slots][2]).  A workgroup owns whole sequences x one head and keeps their q, k, v in LDS - they never reach HBM.  Rows:
 * nseq sequences of L tokens (row = s * L + t), or - seq_off / seq_len both non-null (device arrays) - seq_len[s] <= L tokens
 * from row seq_off[s], packed back to back; m_dev (device, may be null) = count of valid rows.  W = heads * 64, L <= 256.
 * L <= 56: every operand of a (sequence, 16-query tile) item in registers, bit-identical to cc_linear_ln_f16 followed by
 * cc_attention_f16.  56 < L <= 256 (round 5: ViT-B/16's 197-token frames, its 101 / 161-token clustered blocks, CLIP's
 * 77-token captions): scores and P stay in registers, the PV contraction takes the keys of a 32-key block in the accumulator's
 * own order - equal to the two launches to the rounding of the fp16 output.  CC_ERR_UNSUPPORTED outside that range. */
int cc_inproj_attention_f16(const void* h_f16, const void* w_ln_f16, const float* c1, const float* c2, const float* stats,
                            int32_t slots, float eps, void* att_f16, int32_t nseq, int32_t L, int32_t heads, int32_t causal,
                            const int32_t* seq_off, const int32_t* seq_len, const int32_t* m_dev, void* stream);
/* host-side query: the tile (1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64, 5 = 256x256, 6 = 256x128, 7 = 256x192, 8 = 64x64 with
 * 128-deep k-steps, 10 = 128x256) the dispatcher picks for this shape / epilogue id (CC_EPI_*; 5, 6 =
 * LN-folded f16 without / with QuickGELU, 7 = residual + statistics); <= 0: unsupported */
int cc_linear_tile_for(int32_t M, int32_t N, int32_t K, int32_t epilogue);
/* host-side query: slots per row cc_linear_resid_stats_f16 will write for this shape (tile 0 = auto); <= 0: unsupported */
int cc_linear_resid_stats_slots(int32_t M, int32_t N, int32_t K, int32_t tile);
int cc_linear_resid_stats_f16(const void* a_f16, const void* w_f16, const float* bias, float* h,
                              void* h16_out, float* stats_out, int32_t* slots_out, const float* shift_in,
                              const float* stats_in, int32_t slots_in, float* shift_out,
                              int32_t M, int32_t N, int32_t K, int32_t tile, void* stream);
/* Multi-head self-attention core of nn.MultiheadAttention (modules/clip.py:220-226):
 * qkv [nseq*L, 3W] fp16 (row = seq*L + token; q | k | v, heads = contiguous 64-wide slices),
 * out [nseq*L, W] fp16 = softmax(q k^T / 8 + mask) v; causal != 0 adds the strict upper
 * triangular -inf mask of clip.py:448-454.  head_dim is 64 (W == 64*heads), L <= 256. */
int cc_attention_f16(const void* qkv_f16, void* out_f16, int32_t nseq, int32_t L, int32_t heads,
                     int32_t W, int32_t causal, void* stream);
/* The same for other row orders: token t of sequence s is row s*seq_rows + t*tok_rows of qkv / out.  The LND
 * activations ResidualAttentionBlock.forward receives (modules/clip.py:228-253: x [L, N, W]) are seq_rows = 1,
 * tok_rows = N; cc_attention_f16 is seq_rows = L, tok_rows = 1. */
int cc_attention_strided_f16(const void* qkv_f16, void* out_f16, int32_t nseq, int32_t L, int32_t heads,
                             int32_t W, int32_t causal, int64_t seq_rows, int64_t tok_rows, void* stream);

/* One ResidualAttentionBlock (state-dict keys resblocks.{i}.*) */
typedef struct cc_block_weights {
    const float* ln_1_weight;  const float* ln_1_bias;
    const void* in_proj_weight_f16;   /* [3W, W] */   const float* in_proj_bias;    /* [3W] */
    const void* out_proj_weight_f16;  /* [W, W]  */   const float* out_proj_bias;   /* [W]  */
    const float* ln_2_weight;  const float* ln_2_bias;
    const void* c_fc_weight_f16;      /* [4W, W] */   const float* c_fc_bias;       /* [4W] */
    const void* c_proj_weight_f16;    /* [W, 4W] */   const float* c_proj_bias;     /* [W]  */
    /* ln_1 / ln_2 folded into the Linear layer that consumes them (cc_fold_layernorm_linear_f32):
     *   LN(h) W^T + b = rstd (h (W*gamma)^T - mu c1) + c2
     * so the encoders never run a stand-alone LayerNorm pass over the residual stream.           */
    const void* in_proj_ln_weight_f16; /* fp16(in_proj_weight * ln_1.weight)  [3W, W] */
    const float* in_proj_ln_c1;        /* [3W] row sums of the folded weight          */
    const float* in_proj_ln_c2;        /* [3W] in_proj_weight ln_1.bias + in_proj_bias */
    const void* c_fc_ln_weight_f16;    /* fp16(c_fc.weight * ln_2.weight)     [4W, W] */
    const float* c_fc_ln_c1;           /* [4W] */
    const float* c_fc_ln_c2;           /* [4W] */
} cc_block_weights;

/* Fold a LayerNorm (gamma, beta over K) into the Linear layer y = LN(x) W^T + b that consumes it:
 * w_f16_out[n,k] = fp16(W[n,k] * gamma[k]); c1_out[n] = sum_k w_f16_out[n,k]; c2_out[n] = sum_k beta[k] W[n,k] + b[n].
 * weight [N,K] fp32, bias [N] fp32 or NULL.  Fills the *_ln_* fields of cc_block_weights. */
int cc_fold_layernorm_linear_f32(const float* weight, const float* bias, const float* gamma, const float* beta,
                                 int32_t N, int32_t K, void* w_f16_out, float* c1_out, float* c2_out, void* stream);

#define CC_MAX_LAYERS 32

/* cc_vit_model.row_policy / cc_text_model.row_policy: compute rows whose values nothing downstream reads, exactly as the
 * reference does (bench.py and the tests time / compare both forms; results agree bit for bit for the text rows and to the
 * rounding of the fp16 intermediates for the last block). */
#define CC_ROWS_ALL_TEXT        1   /* text tower: every token row, not only those up to each caption's EOT            */
#define CC_ROWS_ALL_LAST_BLOCK  2   /* last block of the tower: every row behind the attention, not only CLS / EOT rows */

/* VisualTransformer + the ln_post/proj tail of CLIP.encode_image (modules/clip.py:272-349,460-469) */
typedef struct cc_vit_model {
    int32_t layers, width, heads, patch, resolution, embed_dim;
    const void* conv1_weight_f16;         /* [W, 3*p*p]                         */
    const float* class_embedding;         /* [W]                                */
    const float* positional_embedding;    /* [1 + (res/p)^2, W]                 */
    const float* ln_pre_weight;  const float* ln_pre_bias;
    const float* ln_post_weight; const float* ln_post_bias;
    const float* proj;                    /* [W, embed_dim] fp32                */
    const cc_block_weights* blocks;       /* HOST array [layers]                */
    /* token-cluster plan, one entry per block (the per-block decision of get_cluster_inter,
     * modules/cluster/cluster.py:15-37): cluster_tokens[i] > 0 => before the attention of block i
     * (0-based) the clip's frames are cut to cluster_frames[i] segments of cluster_tokens[i]
     * medoid tokens (clip.py:236-242). */
    int32_t cluster_frames[CC_MAX_LAYERS];
    int32_t cluster_tokens[CC_MAX_LAYERS];
    int32_t cluster_metric;   float cluster_norm_p;   float cluster_threshold;
    int32_t cluster_iter_limit, cluster_split_size, cluster_pre_norm;
    /* optional HOST array [layers]: the TokenClusterInter variant of block i (N2); NULL = medoid gather
     * everywhere.  For CC_CLUSTER_POOLING cluster_tokens[i] must equal the incoming token count. */
    const struct cc_cluster_variant* cluster_variants;
    /* CC_ROWS_* bits; 0 = the shipped policy (the last block computes out_proj / c_fc / c_proj for the rows the
     * projection head reads).  Per model, not process state: two models with different policies can run side by side. */
    int32_t row_policy;
    /* linear_patch = '3d' (modules/clip.py:296-317): the patch embedding is a Conv3d over (t, h, w) with kernel (3, p, p),
     * stride (1, p, p) and zero padding 1 along t - frame t of a clip sees frames t-1, t, t+1 of the SAME clip.
     * conv2_weight_f16 [W, 3*3*p*p] = the Conv3d weight [W, 3(c), 3(t), p, p] flattened; NULL = '2d' (conv1). */
    const void* conv2_weight_f16;
} cc_vit_model;

/* Frame input descriptor for the *_frames entry points.  The reference's evaluation
 * loader turns a decoded uint8 frame into the float tensor the ViT eats with three fp32 ops -
 * x = u8 / 255 (dataloaders/transforms.py:166, GroupToTensorBCHW(div=True)), then x = (x - mean) / std
 * (transforms.py:19-34 -> torchvision normalize: sub_, div_) with the constants of dataloaders/decode.py:43-48.
 * Handing over the uint8 frames lets the patch gather do exactly these three IEEE fp32 operations itself
 * (bit-identical patches), reading 1 byte per sample from HBM (and over PCIe) instead of 4. */
#define CC_FRAMES_F32_CHW 0   /* [F, 3, res, res] fp32, already normalised (what CLIP.encode_image takes)   */
#define CC_FRAMES_U8_CHW  1   /* [F, 3, res, res] uint8                                                   */
#define CC_FRAMES_U8_HWC  2   /* [F, res, res, 3] uint8 (decoder layout, before transforms.py:157 permutes) */
typedef struct cc_frames {
    const void* data;
    int32_t format;           /* CC_FRAMES_*                                         */
    float mean[3], std[3];    /* per channel; used by the uint8 formats only         */
} cc_frames;

size_t cc_vit_workspace_bytes(const cc_vit_model* m, int32_t B, int32_t T);

/* Number of int64 ids the forced_medoids argument of cc_vit_encode* / cc_clip_encode* must hold for a batch of B clips: the
 * sum over the tower's cluster blocks of B x cluster_frames[i] x cluster_tokens[i] (0: no cluster block, < 0: bad arguments).
 * The pointer carries no length - a host-side caller checks its buffer against this before the call. */
int64_t cc_vit_forced_medoids_count(const cc_vit_model* m, int32_t B);

/* CLIP.encode_image(image, video_frame=T) (modules/clip.py:460-469): video [B*T, 3, res, res]
 * fp32 -> features [B*T_final, embed_dim] fp32 (CLS row of ln_post(hidden) @ proj; only the
 * CLS row is projected - identical values).  hidden_out (optional):
 * the final hidden state [B*T_final, L_final, W] fp32 before ln_post (VisualTransformer.forward
 * output, clip.py:304-349).  medoids_out (optional): int64 [T_new*B, K] of the LAST k-medoids block of the
 * plan (T_new, K of THAT block; earlier cluster blocks do not write it).
 * forced_medoids (optional, test hook for "embeddings given identical medoid sets"):
 * int64 [T_new*B, K] per cluster block, the blocks' tensors back to back in block order; when non-NULL every cluster block
 * (shipped variant: k-medoids, medoid aggregation) skips the selection and gathers these ids instead. */
int cc_vit_encode(const cc_vit_model* m, const float* video, int32_t B, int32_t T,
                  float* features, float* hidden_out, int64_t* medoids_out,
                  const int64_t* forced_medoids, void* ws, size_t ws_bytes, void* stream);

/* Text transformer + ln_final/text_projection tail of CLIP.encode_text (modules/clip.py:471-496) */
typedef struct cc_text_model {
    int32_t layers, width, heads, context_length, vocab_size, embed_dim;
    const float* token_embedding;         /* [vocab, W]        */
    const float* positional_embedding;    /* [context, W]      */
    const float* ln_final_weight; const float* ln_final_bias;
    const float* text_projection;         /* [W, embed_dim]    */
    const cc_block_weights* blocks;       /* HOST array [layers] */
    int32_t row_policy;                   /* CC_ROWS_* bits; 0 = shipped policy */
} cc_text_model;

size_t cc_text_workspace_bytes(const cc_text_model* m, int32_t Bt, int32_t Lt);

/* ids [Bt, Lt] int64 -> features [Bt, embed_dim] fp32: row at the first argmax of the ids
 * (EOT has the largest id, clip.py:484) of ln_final(x) @ text_projection.
 * Caption compaction: the attention mask is causal (clip.py:448-454) and only the EOT row is projected, so tokens behind
 * a caption's EOT cannot reach its feature.  The tower therefore runs on the rows up to each caption's EOT only, packed
 * back to back (positions found on the device, launches sized for Bt*Lt rows, no host synchronisation) - the features
 * are bit-identical to running all Bt*Lt rows.  cc_text_encode_hidden with hidden_out != NULL runs every row. */
int cc_text_encode(const cc_text_model* m, const int64_t* ids, int32_t Bt, int32_t Lt,
                   float* features, void* ws, size_t ws_bytes, void* stream);
/* ... also returning the final hidden state [Bt, Lt, W] fp32 before ln_final (hidden_out may be NULL) */
int cc_text_encode_hidden(const cc_text_model* m, const int64_t* ids, int32_t Bt, int32_t Lt,
                          float* features, float* hidden_out, void* ws, size_t ws_bytes, void* stream);

/* out[r] = LayerNorm(h[r*row_mul + (row_idx ? row_idx[r] : 0)]) @ proj   (fp32 throughout; h rows of W floats,
 * proj [W, E], out [R, E]).  The ln_post @ proj / ln_final @ text_projection tail of CLIP.encode_image / encode_text
 * (modules/clip.py:463,480) for any set of rows: row_mul = 1, row_idx = NULL projects every token
 * (return_hidden=True, clip.py:466-467,491-492).  W <= 1024, E % 4 == 0, R <= 65535. */
int cc_head_project_f32(const float* h, int32_t row_mul, const int32_t* row_idx, const float* gamma,
                        const float* beta, const float* proj, float* out, int32_t R, int32_t W, int32_t E,
                        void* stream);

/* Both encoders of one CLIP4Clip.forward call (modules/clip4clip.py:199-243: get_sequence_output
 * + get_visual_output) in one enqueue.  Block i of the text tower shares every launch with block i
 * of the ViT (horizontal fusion): same results as cc_vit_encode + cc_text_encode, fewer and fuller
 * launches.  ws: cc_clip_workspace_bytes. */
size_t cc_clip_workspace_bytes(const cc_vit_model* vm, int32_t B, int32_t T,
                               const cc_text_model* tm, int32_t Bt, int32_t Lt);
int cc_clip_encode(const cc_vit_model* vm, const float* video, int32_t B, int32_t T,
                   float* visual_features, int64_t* medoids_out,
                   const cc_text_model* tm, const int64_t* ids, int32_t Bt, int32_t Lt,
                   float* text_features, void* ws, size_t ws_bytes, void* stream);

/* cc_vit_encode / cc_clip_encode with the frames given through a cc_frames descriptor (N3): for
 * CC_FRAMES_F32_CHW identical to the calls above; for the uint8 formats the normalisation of
 * dataloaders/transforms.py is fused into the patch gather.  Same outputs, same workspace. */
int cc_vit_encode_frames(const cc_vit_model* m, const cc_frames* frames, int32_t B, int32_t T,
                         float* features, float* hidden_out, int64_t* medoids_out,
                         const int64_t* forced_medoids, void* ws, size_t ws_bytes, void* stream);
int cc_clip_encode_frames(const cc_vit_model* vm, const cc_frames* frames, int32_t B, int32_t T,
                          float* visual_features, int64_t* medoids_out, const int64_t* forced_medoids,
                          const cc_text_model* tm, const int64_t* ids, int32_t Bt, int32_t Lt,
                          float* text_features, void* ws, size_t ws_bytes, void* stream);

/* S2 - the meanP similarity tail, CLIP4Clip._loose_similarity (modules/clip4clip.py:357-366) with
 * _mean_pooling_for_similarity_visual (:305-316):
 *   v_hat = v/|v| per frame; v_bar = sum_t mask*v_hat / max(sum_t mask, 1 if 0); v_bar /= |v_bar|
 *   t_hat = t/|t|;  logits[Bt,Bv] = exp(logit_scale) * t_hat v_bar^T           (all fp32)
 * visual [Bv, Tn, E] fp32, video_mask [Bv, Tn] int64 (as the reference passes it), text [Bt, E].
 * pooled_out (optional) [Bv, E] receives v_bar.  ws: (Bv + Bt) * E floats. */
size_t cc_similarity_workspace_bytes(int32_t Bt, int32_t Bv, int32_t E);

/* The operands of the similarity GEMM as a by-product of their producers (the evaluation loop, main.py:381-534: features
 * are cached batch by batch and multiplied once at the end).  A plane row = 3E fp16 values, cc_similarity_plane_row_bytes(E):
 *   cc_normalize_rows_planes_f32        rows / |row| (clip4clip.py:361-362) -> planes [R, 3E] (video_side 0: the text operand,
 *                                       1: rows that are pooled video features already) and, optionally, fp32 `out`
 *   cc_video_pool_normalize_planes_f32  clip4clip.py:305-316,357-360 -> video-side planes [Bv, 3E] (+ fp32 `pooled`, optional)
 *   cc_scaled_dot_planes_f32            logits [Bt, Bv] = mult * text . video^T from the planes: ONE GEMM launch, nothing else.
 *                                       The video plane buffer must hold video_rows >= cc_similarity_padded_rows(Bv) rows with
 *                                       the rows behind Bv ZERO (the tiles read whole multiples of their width).
 *   cc_scaled_dot_planes_products_f32   the same with `products` (3, 2 or 1) of the three fp16 products per multiply-add issued:
 *                                       3 = hi.hi + hi.lo + lo.hi (both operands to 22 bits: 2e-6 on a cosine of unit rows - the
 *                                       default everywhere, rank-exact against the reference's fp32 product on its fixtures);
 *                                       2 = fp16(text) x video-to-22-bits; 1 = fp16 x fp16 (measured on 10k x 1k unit rows:
 *                                       bench.py `similarity_10k_x_1k.products`).  All are inside the 1e-3 BASELINE.json's
 *                                       north_star asks of similarities; the GEMM's matrix-core work is products / 3.
 * E % 64 == 0.  Same values as cc_scaled_dot_nt_f32 on the normalised rows. */
size_t cc_similarity_plane_row_bytes(int32_t E);
int32_t cc_similarity_padded_rows(int32_t Bv);
int cc_normalize_rows_planes_f32(const float* in, float* out, void* planes, int32_t video_side, int32_t R, int32_t E,
                                 void* stream);
int cc_video_pool_normalize_planes_f32(const float* visual, const int64_t* video_mask, int32_t Bv, int32_t Tn, int32_t E,
                                       float* pooled, void* planes, void* stream);
int cc_scaled_dot_planes_f32(const void* text_planes, const void* video_planes, int32_t Bt, int32_t Bv,
                             int32_t video_rows, int32_t E, float mult, float* logits, int32_t ldl, void* stream);
int cc_scaled_dot_planes_products_f32(const void* text_planes, const void* video_planes, int32_t Bt, int32_t Bv,
                                      int32_t video_rows, int32_t E, float mult, int32_t products, float* logits,
                                      int32_t ldl, void* stream);
int cc_video_pool_normalize_f32(const float* visual, const int64_t* video_mask, int32_t Bv, int32_t Tn,
                                int32_t E, float* pooled, void* stream);
int cc_loose_similarity_f32(const float* text, const float* visual, const int64_t* video_mask,
                            int32_t Bt, int32_t Bv, int32_t Tn, int32_t E, float logit_scale,
                            float* logits, int32_t ldl, float* pooled_out,
                            void* ws, size_t ws_bytes, void* stream);
/* The same with the mask addressed through element strides (mask[v, t] = video_mask[v*row_stride + t*col_stride]):
 * the segment mask after clustering is every fd-th column of the frame mask (clip4clip.py:436-447), taken as a strided
 * view instead of a gathered copy. */
int cc_loose_similarity_strided_f32(const float* text, const float* visual, const int64_t* video_mask,
                                    int64_t mask_row_stride, int64_t mask_col_stride, int32_t Bt, int32_t Bv,
                                    int32_t Tn, int32_t E, float logit_scale, float* logits, int32_t ldl,
                                    float* pooled_out, void* ws, size_t ws_bytes, void* stream);
/* The same on the packed all-gather buffer of the multi-GPU step (modules/utils.py:47-64 + clip4clip.py:351-355 as
 * ONE collective): videos come in groups of `group` (one record per rank); video v = (g, l) has its frames at
 * visual + g*vis_group_stride + l*Tn*E (floats) and its mask row at video_mask + g*mask_group_stride +
 * l*mask_row_stride (int64 elements).  group >= Bv is the plain layout. */
int cc_loose_similarity_grouped_f32(const float* text, const float* visual, const int64_t* video_mask, int32_t group,
                                    int64_t vis_group_stride, int64_t mask_group_stride, int64_t mask_row_stride,
                                    int64_t mask_col_stride, int32_t Bt, int32_t Bv, int32_t Tn, int32_t E,
                                    float logit_scale, float* logits, int32_t ldl, float* pooled_out,
                                    void* ws, size_t ws_bytes, void* stream);
/* rows [R, E] -> rows / |row|: the text half of _loose_similarity (clip4clip.py:361-362) for the pre-pooled
 * (2-D visual_output) branch */
int cc_normalize_rows_f32(const float* in, float* out, int32_t R, int32_t E, void* stream);
/* logits[Bt,Bv] = mult * a[Bt,E] b[Bv,E]^T for already-normalised rows (|element| <= 1; the kernel splits each fp32
 * operand into two fp16 parts scaled by 2^10 and accumulates hi.hi + hi.lo + lo.hi in fp32 on the fp16 matrix cores -
 * 22-bit operands, fp32-level results; elements beyond +-63 overflow to inf).  E % 64 == 0.
 * ws: cc_similarity_workspace_bytes(Bt, Bv, E) (the split planes). (the sharded eval
 * similarity matrix, main.py:502-534, computed in one launch per row block). */
int cc_scaled_dot_nt_f32(const float* a, const float* b, int32_t Bt, int32_t Bv, int32_t E, float mult,
                         float* logits, int32_t ldl, void* ws, size_t ws_bytes, void* stream);

/* N4, forward values only - CrossEn (modules/losses.py:8-18) in both directions and their mean as CLIP4Clip.forward's
 * training branch forms it (modules/clip4clip.py:245-253): loss3[0] = mean_i -log_softmax(sim[i,:])[i],
 * loss3[1] = the same on sim^T, loss3[2] = (loss3[0] + loss3[1]) / 2.  sim [n,n] fp32 addressed through element strides.
 * ws: 2*n floats.  (No backward: training is out of scope; this is the loss a validation pass reports.) */
int cc_contrastive_loss_f32(const float* sim, int32_t n, int64_t row_stride, int64_t col_stride, float* loss3,
                            void* ws, size_t ws_bytes, void* stream);

/* N4 - the training branch's loss WITH its gradient (modules/clip4clip.py:245-262, 357-366; modules/losses.py:8-18): from the
 * features as the towers return them - text [n, E], visual [n, Tn, E] (per-segment), video_mask [n, Tn] int64 (element
 * strides) - forms S = exp(logit_scale) * normalise(text) . normalise(masked mean of per-frame-normalised visual)^T,
 * loss3 = (CrossEn(S), CrossEn(S^T), their mean) and, for an incoming gradient grad_scale of loss3[2], what torch.autograd
 * yields for it: d_text [n, E], d_visual [n, Tn, E], d_logit_scale [1].  fp32, fixed summation orders (deterministic).
 * The towers have no backward in this library (encoder backward and DDP all-reduce are out of scope):
 * this is the gradient a caller feeds into its own backward of the encoders.  E <= 1024. */
size_t cc_contrastive_grad_workspace_bytes(int32_t n, int32_t Tn, int32_t E);
int cc_contrastive_loss_grad_f32(const float* text, const float* visual, const int64_t* video_mask,
                                 int64_t mask_row_stride, int64_t mask_col_stride, int32_t n, int32_t Tn, int32_t E,
                                 float logit_scale, float grad_scale, float* loss3, float* d_text, float* d_visual,
                                 float* d_logit_scale, void* ws, size_t ws_bytes, void* stream);
/* the same with logit_scale read from device memory when logit_scale_dev != null (the nn.Parameter itself: a training step then
 * neither waits on the host for its value nor bakes it into a captured hipGraph) */
int cc_contrastive_loss_grad_dev_f32(const float* text, const float* visual, const int64_t* video_mask,
                                     int64_t mask_row_stride, int64_t mask_col_stride, int32_t n, int32_t Tn, int32_t E,
                                     float logit_scale, const float* logit_scale_dev, float grad_scale, float* loss3,
                                     float* d_text, float* d_visual, float* d_logit_scale, void* ws, size_t ws_bytes,
                                     void* stream);

/* N1 - the rank extraction of compute_metrics (utils/metrics.py:11-26) on the device: for row i with
 * ground-truth column g = diag_offset + i, counts[2i] = #{j: sim[i,j] > sim[i,g]} and counts[2i+1] =
 * #{j: sim[i,j] == sim[i,g]} (>= 1).  The reference's rank list `ind` is the concatenation over rows of
 * range(counts[2i], counts[2i] + counts[2i+1]) - R@k / MdR / MnR follow from 2 ints per row instead of a
 * device->host copy and NumPy sort of the whole matrix.  Element (i,j) = sim[i*row_stride + j*col_stride]
 * (swap the strides to rank sim^T, i.e. video->text). */
int cc_rank_counts_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                       int32_t diag_offset, int32_t* counts, void* stream);

/* The same with an explicit ground-truth column per row (gt_cols [rows] int32, device) - the multi-sentence
 * protocol of tensor_text_to_video_metrics (utils/metrics.py:38-65), where the sentences of one video share its
 * column.  counts3 [rows,3]: #greater, #equal (>= 1), #equal with a smaller column index (the position of the
 * ground truth among its ties under a stable descending sort; the reference's argsort leaves that order open). */
int cc_rank_counts_cols_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                            const int32_t* gt_cols, int32_t* counts3, void* stream);

/* The same against a reference value handed in per row (ref_vals [rows] fp32, device): counts [rows,2] = #entries of the
 * row greater than / equal to ref_vals[i].  For the clip-sharded evaluation: a rank holds a ROW block
 * [Nt/G, Nv] of the matrix of main.py:502-534; the video->text rank of video v counts, over all text rows, the entries of
 * column v above the ground-truth entry, which lives in one rank's block - every rank counts its rows against the
 * broadcast ground-truth values (swap the strides to walk columns) and the per-rank counts are summed. */
int cc_rank_counts_ref_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                           const float* ref_vals, int32_t* counts, void* stream);

/* tensor_video_to_text_sim (utils/metrics.py:68-76) on the device: out [n_groups, cols] = for every group g and column
 * (video) v the maximum of sim[r, v] over the rows (sentences) r with group[r] == g, NaN entries ignored, -inf where a
 * group has no row here (a rank's row block of the clip-sharded evaluation: the per-rank results are combined with a MAX
 * all-reduce).  group [rows] int32 in [0, n_groups); rows = 0 only fills out with -inf. */
int cc_group_max_rows_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, const int32_t* group,
                          int32_t n_groups, float* out, void* stream);

/* ==========================================================================================
 * N4, first slice of training: the backward of one ResidualAttentionBlock (modules/clip.py:228-253; the reference gets
 * it from torch.autograd in main.py:321).  The dgrad / wgrad contractions run on cc_linear_f16 with swapped operand roles
 * (centerclip_amd/train.py); these entry points are the pieces that are not a GEMM.  fp32 arithmetic, fixed summation
 * orders (identical bits on every run).
 *   cc_layernorm_backward_f32   x [rows, W] (row stride x_stride), dy [rows, W] -> dx [rows, W] = dres (optional, the
 *                               residual branch's gradient) + LayerNorm backward; dgamma, dbeta [W].  W % 4 == 0, W <= 1024.
 *   cc_quick_gelu_backward_f16  du_pre = du * d/dx [x sigmoid(1.702 x)] at x = u_pre (fp16, the c_fc output before the
 *                               activation, clip.py:192-194); n % 4 == 0
 *   cc_attention_backward_f16   qkv [nseq*L, 3W] fp16 (as cc_attention_f16), d_out [nseq*L, W] fp32 -> d_qkv [nseq*L, 3W] fp32
 *                               (softmax(q k^T / 8 + mask) v per 64-wide head); fp16 MFMA operands (dO and dS scaled by
 *                               powers of two chosen on the device), fp32 accumulators.  L <= 64: one launch, a workgroup
 *                               per (sequence, head).  64 < L <= 256 (ViT-B/16): a query-side launch (dQ; log-sum-exp and
 *                               D = sum_j P dP per query into ws) and a key-side launch (dV, dK)
 *   cc_column_sums_f32          out [cols] = column sums of in [rows, cols] (bias gradients)
 *   cc_cast_scaled_f16          fp32 -> fp16 with a power-of-two scale chosen on the device from the tensor's largest
 *                               magnitude (scale * max in [8192, 16384)); *scale_out (device) receives it; amax_scratch: one
 *                               device float.  cc_unscale_f32 divides an fp32 product by one or two such scales.
 * ========================================================================================== */
size_t cc_layernorm_backward_workspace_bytes(int32_t rows, int32_t W);
/* (dx_amax / out_amax below, may be null: one device float that holds 0 - or an earlier maximum - on entry and max(it, the
 *  largest magnitude written) on exit, so that cc_cast_transpose_f16(scaled = 2) can scale that tensor without reading it
 *  first; one atomic per workgroup of the producing kernel) */
int cc_layernorm_backward_f32(const float* x, int64_t x_stride, const float* gamma, const float* dy, const float* dres,
                              float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t W, float eps, float* dx_amax,
                              void* ws, size_t ws_bytes, void* stream);
int cc_quick_gelu_f16(const void* in_f16, void* out_f16, int64_t n, void* stream);     /* QuickGELU on fp16 (training forward) */
int cc_quick_gelu_backward_f16(const void* u_pre_f16, const float* du, float* du_pre, int64_t n, float* out_amax, void* stream);
size_t cc_attention_backward_workspace_bytes(int32_t nseq, int32_t L, int32_t heads);    /* 0 for L <= 64 (ws may be null) */
int cc_attention_backward_f16(const void* qkv_f16, const float* d_out, float* d_qkv, int32_t nseq, int32_t L,
                              int32_t heads, int32_t W, int32_t causal, float* out_amax, void* ws, size_t ws_bytes,
                              void* stream);
size_t cc_column_sums_workspace_bytes(int32_t rows, int32_t cols);
int cc_column_sums_f32(const float* in, int32_t rows, int32_t cols, float* out, void* ws, size_t ws_bytes, void* stream);
int cc_cast_scaled_f16(const float* in, void* out_f16, int64_t n, float* amax_scratch, float* scale_out, void* stream);
int cc_unscale_f32(float* x, int64_t n, const float* scale_a, const float* scale_b, void* stream);
/* c [M, N] fp32 = (a [M, K] w [N, K]^T) / *scale_dev: cc_linear_f16's "f32" form with the unscale above folded into the
 * epilogue (the dgrad / wgrad products of an operand cast by cc_cast_scaled_f16 / cc_cast_transpose_f16; the division by a
 * power of two is exact, so the result equals cc_linear_f16 followed by cc_unscale_f32 bit for bit). */
int cc_linear_unscaled_f16(const void* a_f16, const void* w_f16, float* c, int32_t M, int32_t N, int32_t K,
                           const float* scale_dev, void* stream);
/* The fp16 operand copies a Linear's backward multiplies, from ONE read of the matrix: `in` fp32 [rows, cols] (or in_f16, a
 * saved fp16 activation) -> out_f16 [rows, cols] (may be null) and out_t_f16 [cols, rows_pad] (may be null) = the transpose with zero columns
 * behind `rows` (rows_pad >= rows, a multiple of 64: the contraction of dW = dY^T X; cols % 4 == 0).  scaled != 0: the fp32
 * input is scaled by cc_cast_scaled_f16's device-chosen power of two (amax_scratch: one device float, *scale_out the scale);
 * scaled == 2: *amax_scratch already holds the tensor's largest magnitude (written by the kernel that produced the tensor,
 * see dx_amax / out_amax above) and the pass that finds it is skipped.
 * col_sums (may be null; fp32 input only): the column sums of the unscaled matrix [cols] from the same read - the Linear's bias
 * gradient - via per-tile partials in ws (cc_cast_transpose_colsum_workspace_bytes), added in tile order.  col_sums null with a
 * ws of that size: the partials [rows_pad / 64][cols] are left in ws and nothing is added (cc_wgrad_tn_f16 adds them). */
size_t cc_cast_transpose_colsum_workspace_bytes(int32_t rows_pad, int32_t cols);
int cc_cast_transpose_f16(const float* in, const void* in_f16, void* out_f16, void* out_t_f16, int32_t rows, int32_t cols,
                          int32_t rows_pad, int32_t scaled, float* amax_scratch, float* scale_out, float* col_sums, void* ws,
                          size_t ws_bytes, void* stream);

/* Weight gradient of a Linear layer from the row-major matrices the backward holds - no transposed copies (round 5):
 *   dw [N1, N2] fp32 = (dy^T x) / *scale_dev,   dy [M, N1], x [M, N2] fp16 row-major, the contraction over their rows
 * (main.py:321: torch.autograd's dW = dY^T X of y = x W^T).  N1 % 128 == 0, N2 % 128 == 0.  The row range is cut into slices so
 * that the grid fills the chip; the slices' partial products are added in slice order (bit-stable from run to run).
 * scale_dev (may be null: 1): the power of two dy was multiplied by (cc_cast_transpose_f16 / cc_cast_scaled_f16).
 * col_partial / col_chunks / bias_grad (all or none): the per-tile partial column sums [col_chunks][N1] cc_cast_transpose_f16
 * leaves in its ws when called without col_sums (col_chunks = rows_pad / 64) -> bias_grad [N1], added in the association of
 * cc_cast_transpose_f16's own reduction (same bits), in the launch that adds the slices.
 * ws: cc_wgrad_tn_workspace_bytes(M, N1, N2). */
size_t cc_wgrad_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2);
int cc_wgrad_tn_f16(const void* dy_f16, const void* x_f16, float* dw, int32_t M, int32_t N1, int32_t N2,
                    const float* scale_dev, const float* col_partial, int32_t col_chunks, float* bias_grad, void* ws,
                    size_t ws_bytes, void* stream);
/* One BertAdam step on one parameter tensor (utils/optimization.py:100-170: the optimizer main.py:161-167 builds): grad is
 * clipped in place to max_grad_norm (clip_grad_norm_ on the single tensor; <= 0: no clipping), next_m = b1 m + (1-b1) g,
 * next_v = b2 v + (1-b2) g^2, param -= lr_scheduled * (next_m / (sqrt(next_v) + e) + weight_decay * param); no bias correction.
 * lr_scheduled = lr * schedule(step / t_total, warmup) is host arithmetic (centerclip_amd.train.BertAdam).  All tensors fp32. */
size_t cc_bertadam_workspace_bytes(void);
/* (lr_dev, may be null: the scheduled learning rate read from a device float instead of lr_scheduled - a training step captured
 * into a hipGraph is replayed with the schedule's new value written there first) */
int cc_bertadam_step_f32(float* param, float* grad, float* next_m, float* next_v, int64_t n, float lr_scheduled, float b1,
                         float b2, float e, float weight_decay, float max_grad_norm, const float* lr_dev, void* ws, size_t ws_bytes,
                         void* stream);
/* All of a model's small tensors (biases, LayerNorm weights: n <= CC_BERTADAM_MULTI_MAX_N, the size up to which
 * cc_bertadam_step_f32 itself uses one workgroup) in ONE launch: items_dev = `count` records below in device memory, one
 * workgroup each; per tensor the arithmetic and the bits of cc_bertadam_step_f32 (lr_dev != null: the learning rate is read
 * from that device float, otherwise `lr`).  The records hold device pointers: the caller keeps them valid until the launch
 * has run (centerclip_amd.train.BertAdam stages them through pinned memory). */
#define CC_BERTADAM_MULTI_MAX_N 8192
typedef struct cc_bertadam_item {
    float* param; float* grad; float* next_m; float* next_v;
    const float* lr_dev;
    int32_t n;
    float lr, weight_decay;
    int32_t reserved;
} cc_bertadam_item;                                            /* 56 bytes */
int cc_bertadam_multi_f32(const void* items_dev, int32_t count, float b1, float b2, float e, float max_grad_norm, void* stream);
/* ... and of `count` LARGE tensors (any n) in TWO launches - every tensor's norm workgroups, then every tensor's step workgroups,
 * each finding its tensor by bisection: records ordered, norm_blk0 / step_blk0 = running sums of norm_blocks / step_blocks =
 * cc_bertadam_norm_blocks(n) / cc_bertadam_step_blocks(n) (host-side queries), ws >= total_norm_blocks doubles.  Per tensor the
 * arithmetic and the bits of cc_bertadam_step_f32.  (A ViT-B/32 CLIP has ~100 such tensors: 204 launches -> 2.) */
typedef struct cc_bertadam_big_item {
    float* param; float* grad; float* next_m; float* next_v;
    const float* lr_dev;
    int64_t n;
    float lr, weight_decay;
    int32_t norm_blk0, norm_blocks, step_blk0, step_blocks;
} cc_bertadam_big_item;                                        /* 72 bytes */
int32_t cc_bertadam_norm_blocks(int64_t n);
int32_t cc_bertadam_step_blocks(int64_t n);
int cc_bertadam_multi_large_f32(const void* items_dev, int32_t count, int32_t total_norm_blocks, int32_t total_step_blocks,
                                float b1, float b2, float e, float max_grad_norm, void* ws, size_t ws_bytes, void* stream);

/* ==========================================================================================
 * Diagnostics (not on the product path; process-wide state, not thread-safe).
 * While armed, every launch of the tiled GEMM kernel is issued with a start / stop event pair that receives the dispatch's
 * own begin / end timestamps (what rocprofv3 --kernel-trace reads), so a kernel symbol can be timed IN SITU, inside an
 * eagerly enqueued step between its real neighbours.  bench.py's `roofline` uses it.  Never arm during graph capture.
 *   cc_debug_gemm_timing_begin(cap)  arm for up to `cap` launches (earlier records are dropped); cap <= 0 disarms + frees
 *   cc_debug_gemm_timing_end()       disarm; -> number of launches recorded
 *   cc_debug_gemm_timing_read(i, us_out, info12_out)   (after synchronising the stream) duration of launch i in
 *        microseconds + its record: BM, BN, WM, WN, epilogue id, BK | split << 16, then M, N, K of the carrier and of the
 *        rider problem (zeros without a rider).
 * The per-workgroup stamp hooks (cc_debug_set_*_profile) exist only in development builds (-DCC_DEV_KNOBS).
 * ========================================================================================== */
int cc_debug_gemm_timing_begin(int cap);
int cc_debug_gemm_timing_end(void);
int cc_debug_gemm_timing_read(int i, float* us_out, int* info12_out);

#ifdef __cplusplus
}
#endif
#endif /* CENTERCLIP_HIP_H */
