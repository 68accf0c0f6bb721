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
 *      batch_fast_kmedoids (:45-97; pass split_size >= P).  `threshold` is accepted for
 *      signature parity; the stop test is the equivalent fixed-point test (see above).
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

#ifdef __cplusplus
}
#endif
#endif /* CENTERCLIP_HIP_H */
