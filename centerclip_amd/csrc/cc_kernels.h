// Internal launch helpers shared between translation units (not part of the public C ABI).
#pragma once
#include "cc_common.h"

// EPI_*_LN: the A operand is the raw fp16 copy of the residual stream and LayerNorm is folded into the GEMM:
//   LN(h) W^T + b = rstd (h (W*gamma)^T - mu c1) + c2,  c1[n] = sum_k (W*gamma)[n,k],  c2 = W beta + b
// with the row statistics (mu, rstd) reduced from per-tile partial sums the producing epilogue wrote.
// EPI_F32_RESID_STATS: residual add that also emits the fp16 copy of the new rows and those partial sums.
enum { EPI_F16 = 0, EPI_F16_GELU = 1, EPI_F32_RESID = 2, EPI_F32_PATCH = 3, EPI_F32 = 4,
       EPI_F16_LN = 5, EPI_F16_GELU_LN = 6, EPI_F32_RESID_STATS = 7,
       // in_proj with the LayerNorm folded AND the attention of its sequences in the epilogue (cc_gemm_attn_dispatch2 only):
       // the q, k, v rows of a tile never leave the CU - C is the attention output [M, W] fp16
       EPI_ATTN_LN = 8 };
#define CC_LN_MAX_SLOTS 32

struct GemmArgs {
    const _Float16* A;   // [M, K] fp16 row-major
    const _Float16* W;   // [N, K] fp16 (nn.Linear layout)
    const float* bias;   // [N] or null
    void* C;             // fp16 or fp32, row stride ldc
    const float* R;      // residual epilogues: the rows that are added (row stride ldc); null = C (in place)
    const float* pos;    // EPI_F32_PATCH: positional embedding [1+n, N]
    int M, N, K, ldc;
    int lda, ldw;        // row strides of A / W in halfs; 0 = K (gemm_f16_kernel only: the similarity GEMM reads the first
                         // `products` of three planes per row, so its rows are 3E halfs apart while K = products * E)
    int patch_n;         // EPI_F32_PATCH: patches per frame (out row = f*(n+1) + 1 + i)
    int tiles_m, tiles_n;
    // LayerNorm folding
    const float* ln_stats;   // *_LN: [M][ln_slots][2] partial (sum, sum of squares) of the A rows
    const float* ln_c1;      // *_LN: [N]  (c2 travels in `bias`)
    int ln_slots;
    float ln_eps;
    float* stats_out;        // RESID_STATS: [M][tiles_n * WN][2]
    _Float16* c16;           // RESID_STATS: fp16 copy of the updated rows (row stride ldc)
    // RESID_STATS row centring: the fp16 copy is fp16(h - c_row) and the statistics are those of the centred copy
    // (LayerNorm is invariant to a per-row shift, so the consuming *_LN GEMM needs no change); c_row = the row mean one
    // sublayer ago = shift_in[m] + mean of the centred copy the previous sublayer wrote (its partial sums
    // shift_stats [M][shift_slots][2]); written to shift_out [M].  All null: c_row = 0.
    const float* shift_in;
    const float* shift_stats;
    int shift_slots;
    float* shift_out;
    // Device-side row count (text tower with compacted captions): when non-null the kernel reads the number of valid
    // rows from *m_dev (<= M, which sizes the grid); tiles beyond it exit at once, rows beyond it are never stored.
    const int* m_dev;
    // Row selection (cc_gemm_rows_dispatch2 only): logical row m of the problem lives at physical row
    // prow(m) = row_map ? row_map[m] : m * row_step (row_step 0 = 1) of A / C / c16 and of every per-row side array
    // (statistics, shifts).  The last block of a tower only needs the rows the projection head reads (the CLS token of
    // every frame: row_step = tokens per frame; the EOT token of every caption: row_map), so its out_proj / c_fc /
    // c_proj run on those rows in place.
    int row_step;
    const int* row_map;
    // EPI_F32 only: C = out_scale * (A W^T + bias) (0 = 1), and only columns n < n_valid are stored (0 = N; N itself must
    // be a multiple of the tile width, so W carries rows up to N - their products are computed and dropped); ldc may be
    // any value >= n_valid (unaligned rows fall back to scalar stores).  The similarity GEMM of similarity.hip.
    float out_scale;
    int n_valid;
    // EPI_F32 only: when non-null the product is also divided by *out_unscale_dev (a power-of-two operand scale chosen on the
    // device by cc_cast_scaled_f16 / cc_cast_transpose_f16: the backward's dgrad / wgrad GEMMs undo it in their epilogue).
    const float* out_unscale_dev;
    // EPI_ATTN_LN (cc_gemm_attn_dispatch2): N = 3W, W = heads * 64; a tile is att_spt whole sequences (rows) x one head
    // (its 64 q, 64 k and 64 v columns); sequence s = att_L tokens at row s * att_L, or - compacted captions -
    // att_seq_len[s] tokens at row att_seq_off[s] (att_L is then the upper bound)
    int att_L, att_nseq, att_spt, att_causal;
    const int* att_seq_off;
    const int* att_seq_len;
};

// Up to two independent GEMM problems with the same epilogue in ONE launch (horizontal fusion of the
// visual and the text tower: the small text problem rides in the tail round of the large one).
struct GemmPair {
    GemmArgs p[2];
    int tiles0;          // workgroups [0, tiles0) -> p[0], the rest -> p[1]
    int rider_prio;      // raise the wave priority of p[1]'s workgroups
};

int cc_gemm_dispatch(GemmArgs g, int epi, int tile, hipStream_t st);
// slots_out (optional, [2]): for RESID_STATS the number of partial-sum slots per row each problem wrote
int cc_gemm_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* slots_out = nullptr);
// The same product for a FEW selected rows (GemmArgs::row_step / row_map; epilogues EPI_F16_GELU_LN, EPI_F32_RESID and
// EPI_F32_RESID_STATS; N % 32 == 0, K % 32 == 0; with statistics N <= 32 * CC_LN_MAX_SLOTS): latency-bound, so the K
// range is split over the 8 waves of a workgroup and the grid has one workgroup per 32 output columns.
// diagnostics (cc_debug_gemm_timing_*): while armed, -> true and a start / stop event pair for this launch
bool cc_gemm_timing_claim(const int rec12[12], hipEvent_t* start, hipEvent_t* stop);
bool cc_gemm_rows_ok(int N, int K, int epi);
int cc_gemm_rows_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, hipStream_t st, int* slots_out = nullptr);

struct LnArgs {
    const float* in; int64_t in_stride; const float* gamma; const float* beta; void* out; int64_t out_stride;
    int rows, W;
    _Float16* out16;     // optional (fp32-output variant): fp16 copy of the output rows MINUS their mean, row stride W
    float* stats;        // optional: [rows][2] (sum, sum of squares) of that centred fp16 copy (one slot)
    float* shift;        // optional: [rows] the mean that was subtracted
    const float* cls;    // optional (ln_pre): rows with row % cls_period == 0 take cls + pos0 as their input
    const float* pos0;
    int cls_period;
};
struct TextEmbedArgs {
    const long long* ids; const float* tok_emb; const float* pos; float* h; int* eot; int Bt, Lt, W;
    _Float16* h16;       // optional: centred fp16 copy of the rows + its (sum, sum of squares) [rows][2] + the row means
    float* stats;
    float* shift;
    // Caption compaction: tokens behind a caption's EOT cannot reach its feature (causal attention, and only the EOT row is
    // projected, modules/clip.py:480-484), so only rows t <= eot[b] are kept, packed back to back: row seq_off[b] + t.
    // seq_off / seq_len [Bt], m_total [1] = number of kept rows, eot[b] = ABSOLUTE row of caption b's EOT token.
    // All null: every row of the [Bt, Lt] grid is kept at b*Lt + t and eot[b] is the position inside the caption.
    int* seq_off;
    int* seq_len;
    int* m_total;
    // rows of tok_emb; 0 = unchecked.  Ids outside [0, vocab) are clamped instead of read out of bounds (nn.Embedding raises
    // for them; an enqueue-only C entry point cannot, and a wild read takes the process down with a memory fault).
    int vocab;
};
// fp16 copy + (sum, sumsq) of fp32 rows (one wave per row); rows contiguous with stride W
int cc_launch_row_stats(const float* h, _Float16* h16, float* stats, float* shift, int rows, int W, hipStream_t st);
int cc_launch_layernorm2(const LnArgs& a0, const LnArgs* a1, float eps, int out_f16, hipStream_t st);

struct AttArgs {
    const _Float16* qkv; _Float16* out; int nseq, L, heads, W, causal;
    // row of (sequence s, token t) in qkv / out = s*seq_rows + t*tok_rows; 0, 0 = frame-major (seq_rows = L, tok_rows = 1)
    int64_t seq_rows, tok_rows;
    // variable-length sequences packed back to back (compacted captions): sequence s has seq_len[s] tokens starting at
    // row seq_off[s] of qkv / out (L is then the upper bound that sizes the kernel); null = nseq sequences of L tokens
    const int* seq_off;
    const int* seq_len;
};
int cc_launch_attention2(const AttArgs& a0, const AttArgs* a1, hipStream_t st);
// in_proj (LayerNorm folded) + attention in one launch (gemm.hip, EPI_ATTN_LN): g.A = centred fp16 rows, g.W / bias / ln_* as
// for EPI_F16_LN, g.C = attention output [M, W] fp16; the att_* fields describe the sequences.  applies(): head width 64,
// att_L <= 256 and the folded form available - otherwise the caller runs the two launches.
bool cc_gemm_attn_applies(const GemmArgs& g0, const GemmArgs* g1);
int cc_gemm_attn_dispatch2(GemmArgs g0, const GemmArgs* g1, hipStream_t st);

int cc_launch_im2col(const cc_frames& frames, _Float16* A, int F, int res, int p, hipStream_t st);
int cc_launch_im2col3d(const cc_frames& frames, _Float16* A, int F, int T, int res, int p, hipStream_t st);   // linear_patch '3d'

// eig.hip: direct symmetric eigensolver for the K smallest eigenpairs (N <= 196, K <= 64, 2K <= N), see the file header
bool cc_sym_eig_tridiag_supports(int N, int K);
size_t cc_sym_eig_tridiag_ws_bytes(int P, int N);
bool cc_sym_eig_tridiag_big_supports(int N, int K);              // 196 < N <= 640, K <= 128: the matrix in a global scratch
size_t cc_sym_eig_tridiag_big_ws_bytes(int P, int N);
int cc_launch_sym_eig_tridiag(const float* laplacian, int P, int N, int K, int correct_sign, float* Q, int ldq, float* evals,
                              int* sweeps_out, void* ws, size_t ws_bytes, hipStream_t st);
int cc_launch_text_embed(const TextEmbedArgs& e, hipStream_t st);
// ln_pre (args as cc_launch_layernorm2, fp32 output) and the text embedding in one launch
int cc_launch_pre_stage(const LnArgs& ln, const TextEmbedArgs& te, float eps, hipStream_t st);
struct HeadArgs {        // out[r] = LN(h[r*row_mul + (row_idx ? row_idx[r] : 0)]) @ proj[W, E]
    const float* h; int row_mul; const int* row_idx; const float* gamma; const float* beta; const float* proj;
    float* out; int R, W, E;
};
int cc_launch_head_project2(const HeadArgs& a0, const HeadArgs* a1, hipStream_t st);
int cc_launch_head_project(const float* h, int row_mul, const int* row_idx, const float* gamma, const float* beta,
                           const float* proj, float* out, int R, int W, int E, hipStream_t st);

// cluster.hip: cc_token_gather_f32 / cc_token_cluster_variant_f32 with by-products for the next block's folded ln_1
// (row_h16 [rows][W] fp16 copy of the dense output rows, row_stats [rows][2] their (sum, sum of squares))
#define CC_INTERNAL __attribute__((visibility("hidden")))      /* shared between translation units, not exported */
extern "C" {
CC_INTERNAL int cc_token_gather_rows(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                         int32_t T_new, int32_t n, int32_t W, int32_t K, const int64_t* medoids, float* out,
                         int64_t out_tok_stride, int64_t out_frame_stride, _Float16* row_h16, float* row_stats,
                         float* row_shift, void* stream);
CC_INTERNAL int cc_token_cluster_variant_rows(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                                  int32_t T_new, int32_t n, int32_t W, int32_t K, int32_t metric, float norm_p,
                                  float threshold, int32_t iter_limit, int32_t split_size, int32_t pre_norm,
                                  const cc_cluster_variant* var, float* out, int64_t out_tok_stride,
                                  int64_t out_frame_stride, int64_t* medoids, int64_t* assign, int32_t* iters, void* ws,
                                  size_t ws_bytes, _Float16* row_h16, float* row_stats, float* row_shift, void* stream);
}
