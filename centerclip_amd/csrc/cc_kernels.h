// Internal launch helpers shared between translation units (not part of the public C ABI).
#pragma once
#include "cc_common.h"

// EPI_*_LN: the A operand is the raw fp16 copy of the residual stream and LayerNorm is folded into the GEMM:
//   LN(h) W^T + b = rstd (h (W*gamma)^T - mu c1) + c2,  c1[n] = sum_k (W*gamma)[n,k],  c2 = W beta + b
// with the row statistics (mu, rstd) reduced from per-tile partial sums the producing epilogue wrote.
// EPI_F32_RESID_STATS: residual add that also emits the fp16 copy of the new rows and those partial sums.
enum { EPI_F16 = 0, EPI_F16_GELU = 1, EPI_F32_RESID = 2, EPI_F32_PATCH = 3, EPI_F32 = 4,
       EPI_F16_LN = 5, EPI_F16_GELU_LN = 6, EPI_F32_RESID_STATS = 7 };
#define CC_LN_MAX_SLOTS 32

struct GemmArgs {
    const _Float16* A;   // [M, K] fp16 row-major
    const _Float16* W;   // [N, K] fp16 (nn.Linear layout)
    const float* bias;   // [N] or null
    void* C;             // fp16 or fp32, row stride ldc
    const float* pos;    // EPI_F32_PATCH: positional embedding [1+n, N]
    int M, N, K, ldc;
    int patch_n;         // EPI_F32_PATCH: patches per frame (out row = f*(n+1) + 1 + i)
    int tiles_m, tiles_n;
    // LayerNorm folding
    const float* ln_stats;   // *_LN: [M][ln_slots][2] partial (sum, sum of squares) of the A rows
    const float* ln_c1;      // *_LN: [N]  (c2 travels in `bias`)
    int ln_slots;
    float ln_eps;
    float* stats_out;        // RESID_STATS: [M][tiles_n * WN][2]
    _Float16* c16;           // RESID_STATS: fp16 copy of the updated rows (row stride ldc)
};

// Up to two independent GEMM problems with the same epilogue in ONE launch (horizontal fusion of the
// visual and the text tower: the small text problem rides in the tail round of the large one).
struct GemmPair {
    GemmArgs p[2];
    int tiles0;          // workgroups [0, tiles0) -> p[0], the rest -> p[1]
};

int cc_gemm_dispatch(GemmArgs g, int epi, int tile, hipStream_t st);
// slots_out (optional, [2]): for RESID_STATS the number of partial-sum slots per row each problem wrote
int cc_gemm_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* slots_out = nullptr);

struct LnArgs {
    const float* in; int64_t in_stride; const float* gamma; const float* beta; void* out; int64_t out_stride;
    int rows, W;
    _Float16* out16;     // optional (fp32-output variant): fp16 copy of the output rows, row stride W
    float* stats;        // optional: [rows][2] (sum, sum of squares) of the OUTPUT rows (one slot)
};
// fp16 copy + (sum, sumsq) of fp32 rows (one wave per row); rows contiguous with stride W
int cc_launch_row_stats(const float* h, _Float16* h16, float* stats, int rows, int W, hipStream_t st);
int cc_launch_layernorm2(const LnArgs& a0, const LnArgs* a1, float eps, int out_f16, hipStream_t st);

struct AttArgs {
    const _Float16* qkv; _Float16* out; int nseq, L, heads, W, causal;
};
int cc_launch_attention2(const AttArgs& a0, const AttArgs* a1, hipStream_t st);

int cc_launch_im2col(const cc_frames& frames, _Float16* A, int F, int res, int p, hipStream_t st);
int cc_launch_cls_pos(float* h, const float* cls, const float* pos, int F, int Ltok, int W, hipStream_t st);
int cc_launch_text_embed(const long long* ids, const float* tok_emb, const float* pos, float* h, int* eot, int Bt,
                         int Lt, int W, hipStream_t st);
int cc_launch_head_project(const float* h, int row_mul, const int* row_idx, const float* gamma, const float* beta,
                           const float* proj, float* out, int R, int W, int E, hipStream_t st);
