// Internal launch helpers shared between translation units (not part of the public C ABI).
#pragma once
#include "cc_common.h"

enum { EPI_F16 = 0, EPI_F16_GELU = 1, EPI_F32_RESID = 2, EPI_F32_PATCH = 3, EPI_F32 = 4 };

struct GemmArgs {
    const _Float16* A;   // [M, K] fp16 row-major
    const _Float16* W;   // [N, K] fp16 (nn.Linear layout)
    const float* bias;   // [N] or null
    void* C;             // fp16 or fp32, row stride ldc
    const float* pos;    // EPI_F32_PATCH: positional embedding [1+n, N]
    int M, N, K, ldc;
    int patch_n;         // EPI_F32_PATCH: patches per frame (out row = f*(n+1) + 1 + i)
    int tiles_m, tiles_n;
};

// Up to two independent GEMM problems with the same epilogue in ONE launch (horizontal fusion of the
// visual and the text tower: the small text problem rides in the tail round of the large one).
struct GemmPair {
    GemmArgs p[2];
    int tiles0;          // workgroups [0, tiles0) -> p[0], the rest -> p[1]
};

int cc_gemm_dispatch(GemmArgs g, int epi, int tile, hipStream_t st);
int cc_gemm_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, int tile, hipStream_t st);

struct LnArgs {
    const float* in; int64_t in_stride; const float* gamma; const float* beta; void* out; int64_t out_stride;
    int rows, W;
};
int cc_launch_layernorm2(const LnArgs& a0, const LnArgs* a1, float eps, int out_f16, hipStream_t st);

struct AttArgs {
    const _Float16* qkv; _Float16* out; int nseq, L, heads, W, causal;
};
int cc_launch_attention2(const AttArgs& a0, const AttArgs* a1, hipStream_t st);

int cc_launch_im2col(const float* video, _Float16* A, int F, int res, int p, hipStream_t st);
int cc_launch_cls_pos(float* h, const float* cls, const float* pos, int F, int Ltok, int W, hipStream_t st);
int cc_launch_text_embed(const long long* ids, const float* tok_emb, const float* pos, float* h, int* eot, int Bt,
                         int Lt, int W, hipStream_t st);
int cc_launch_head_project(const float* h, int row_mul, const int* row_idx, const float* gamma, const float* beta,
                           const float* proj, float* out, int R, int W, int E, hipStream_t st);
