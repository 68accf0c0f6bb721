// Retrieval tail on gfx950: per-frame L2 normalise -> masked mean over the
// temporal segments -> L2 normalise (CLIP4Clip._mean_pooling_for_similarity_visual + _loose_similarity,
// modules/clip4clip.py:305-316,357-366), and the text x video cosine-logit matrix as ONE exact-fp32
// MFMA "NT" GEMM per row block instead of the reference's batch_size_val^2 tiny matmuls with a
// device->host copy each (main.py:502-534).
#include "cc_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows [R, E] -> rows / |row|   (one wave per row)
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int R, int E) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* src = in + (int64_t)r * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
    const float nrm = sqrtf(cc_wave_sum(s));
    for (int e = lane; e < E; e += 64) out[(int64_t)r * E + e] = src[e] / nrm;
}

// visual [Bv, Tn, E], mask [Bv, Tn] int64 -> pooled [Bv, E]   (one wave per video)
// Videos may come in groups of `vg` (the packed all-gather buffer: one record per rank): video v = (group, local) lives at
// visual + group*vgs + local*Tn*E, its mask row at mask + group*mgs + local*mrs (+ t*mcs).
struct VidAddr {
    int vg;
    int64_t vgs, mgs, mrs, mcs;
};
__global__ __launch_bounds__(256) void video_pool_kernel(const float* __restrict__ visual,
                                                         const long long* __restrict__ mask, VidAddr ad,
                                                         float* __restrict__ pooled, int Bv, int Tn, int E) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= Bv) return;
    const int grp = v / ad.vg, loc = v - grp * ad.vg;
    visual += (int64_t)grp * ad.vgs + (int64_t)loc * Tn * E;
    mask += (int64_t)grp * ad.mgs + (int64_t)loc * ad.mrs;
    constexpr int MAXE = 16;                      // E <= 1024
    float acc[MAXE];
#pragma unroll
    for (int q = 0; q < MAXE; ++q) acc[q] = 0.f;
    float cnt = 0.f;
    for (int t = 0; t < Tn; ++t) {
        const float* src = visual + (int64_t)t * E;
        float x[MAXE];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            x[q] = e < E ? src[e] : 0.f;
            s = fmaf(x[q], x[q], s);
        }
        const float nrm = sqrtf(cc_wave_sum(s));
        const float mk = (float)mask[(int64_t)t * ad.mcs];   // element strides: a strided view is fine
        cnt += mk;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) acc[q] += (x[q] / nrm) * mk;
    }
    if (cnt == 0.f) cnt = 1.f;                    // "avoid zero divide", clip4clip.py:313
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        acc[q] = acc[q] / cnt;
        s = fmaf(acc[q], acc[q], s);
    }
    const float nrm = sqrtf(cc_wave_sum(s));
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = lane + 64 * q;
        if (e < E) pooled[(int64_t)v * E + e] = acc[q] / nrm;
    }
}

// Small batches (a forward step's own Bt x Bv logits): the three launches above are latency-bound (5 + 7 + 14 us for
// 16 x 16), so one workgroup per video does the whole tail: wave 0 pools / normalises the video exactly as
// video_pool_kernel does, parks it in LDS, then the four waves normalise one text each and take the dot product.
__global__ __launch_bounds__(256) void loose_similarity_small_kernel(const float* __restrict__ text,
                                                                     const float* __restrict__ visual,
                                                                     const long long* __restrict__ mask, VidAddr ad,
                                                                     float* __restrict__ logits, int ldl,
                                                                     float* __restrict__ pooled_out, int Bt, int Bv,
                                                                     int Tn, int E, float mult) {
    __shared__ float vp[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int v = blockIdx.x;
    constexpr int MAXE = 16;                      // E <= 1024
    {
        const int grp = v / ad.vg, loc = v - grp * ad.vg;
        visual += (int64_t)grp * ad.vgs + (int64_t)loc * Tn * E;
        mask += (int64_t)grp * ad.mgs + (int64_t)loc * ad.mrs;
    }
    if (wave == 0) {
        float acc[MAXE];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) acc[q] = 0.f;
        float cnt = 0.f;
        for (int t = 0; t < Tn; ++t) {
            const float* src = visual + (int64_t)t * E;
            float x[MAXE];
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const int e = lane + 64 * q;
                x[q] = e < E ? src[e] : 0.f;
                s = fmaf(x[q], x[q], s);
            }
            const float nrm = sqrtf(cc_wave_sum(s));
            const float mk = (float)mask[(int64_t)t * ad.mcs];   // element strides: a strided view is fine
            cnt += mk;
#pragma unroll
            for (int q = 0; q < MAXE; ++q) acc[q] += (x[q] / nrm) * mk;
        }
        if (cnt == 0.f) cnt = 1.f;                // "avoid zero divide", clip4clip.py:313
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            acc[q] = acc[q] / cnt;
            s = fmaf(acc[q], acc[q], s);
        }
        const float nrm = sqrtf(cc_wave_sum(s));
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) {
                const float p = acc[q] / nrm;
                vp[e] = p;
                if (pooled_out) pooled_out[(int64_t)v * E + e] = p;
            }
        }
    }
    __syncthreads();
    for (int t = wave; t < Bt; t += 4) {
        const float* src = text + (int64_t)t * E;
        float x[MAXE];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            x[q] = e < E ? src[e] : 0.f;
            s = fmaf(x[q], x[q], s);
        }
        const float nrm = sqrtf(cc_wave_sum(s));
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) d = fmaf(x[q] / nrm, vp[e], d);
        }
        d = cc_wave_sum(d);
        if (lane == 0) logits[(int64_t)t * ldl + v] = mult * d;
    }
}

// C[i][j] = mult * sum_k A[i][k] B[j][k]; 64x64 tile per workgroup, exact-fp32 MFMA (16x16x4),
// same LDS layout as the Gram kernel of cluster.hip.
#define ST 64
#define SK 32
#define SLD 40
__global__ __launch_bounds__(256) void dot_nt_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                     float* __restrict__ C, int M, int N, int K, int ldc,
                                                     float mult) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][ST * SLD];
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const float* pa[2];
    const float* pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pa[q] = A + (int64_t)min(ti * ST + lrow + 32 * q, M - 1) * K + lchunk * 4;
        pb[q] = B + (int64_t)min(tj * ST + lrow + 32 * q, N - 1) * K + lchunk * 4;
    }
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ra[2], rb[2];
    const int nk = (K + SK - 1) / SK;
    auto gload = [&](int kt) {
        const bool ok = kt * SK + lchunk * 4 < K;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = ok ? *reinterpret_cast<const float4*>(pa[q] + kt * SK) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[q] = ok ? *reinterpret_cast<const float4*>(pb[q] + kt * SK) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<float4*>(&lds[buf][0][(lrow + 32 * q) * SLD + lchunk * 4]) = ra[q];
            *reinterpret_cast<float4*>(&lds[buf][1][(lrow + 32 * q) * SLD + lchunk * 4]) = rb[q];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    const int g = lane >> 4, l15 = lane & 15;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const float* As = &lds[buf][0][(wr * 32 + l15) * SLD + g * 4];
        const float* Bs = &lds[buf][1][(wc * 32 + l15) * SLD + g * 4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(As + ks * 16);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(As + 16 * SLD + ks * 16);
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(Bs + ks * 16);
            const f32x4 b1 = *reinterpret_cast<const f32x4*>(Bs + 16 * SLD + ks * 16);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                // B fragment as the A operand: a lane then owns 4 consecutive j of one i (16-byte stores)
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[tt], a0[tt], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[tt], a0[tt], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[tt], a1[tt], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[tt], a1[tt], acc[1][1], 0, 0, 0);
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int fm = 0; fm < 2; ++fm) {
        const int i = ti * ST + wr * 32 + fm * 16 + l15;
        if (i >= M) continue;
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) {
            const int j = tj * ST + wc * 32 + fn * 16 + g * 4;
            float* dst = C + (int64_t)i * ldc + j;
            const f32x4 v = acc[fm][fn];
            if (j + 3 < N && ((ldc & 3) == 0)) {
                *reinterpret_cast<float4*>(dst) = make_float4(mult * v[0], mult * v[1], mult * v[2], mult * v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < N) dst[e] = mult * v[e];
            }
        }
    }
}

extern "C" {

size_t cc_similarity_workspace_bytes(int32_t Bt, int32_t Bv, int32_t E) {
    if (Bt <= 0 || Bv <= 0 || E <= 0) return 0;
    return cc_align_up((size_t)Bt * E * 4, 256) + cc_align_up((size_t)Bv * E * 4, 256);
}

static int video_pool_launch(const float* visual, const int64_t* video_mask, const VidAddr& ad, int32_t Bv,
                             int32_t Tn, int32_t E, float* pooled, void* stream) {
    if (!visual || !video_mask || !pooled || Bv <= 0 || Tn <= 0 || E <= 0 || ad.vg <= 0) return CC_ERR_INVALID;
    if (E > 1024) return CC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(video_pool_kernel, dim3((Bv + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), visual,
                       reinterpret_cast<const long long*>(video_mask), ad, pooled, Bv, Tn, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_video_pool_normalize_f32(const float* visual, const int64_t* video_mask, int32_t Bv, int32_t Tn, int32_t E,
                                float* pooled, void* stream) {
    const VidAddr ad{Bv > 0 ? Bv : 1, 0, 0, Tn, 1};
    return video_pool_launch(visual, video_mask, ad, Bv, Tn, E, pooled, stream);
}

/* rows [R, E] -> rows / |row| (the text half of _loose_similarity, modules/clip4clip.py:361-362) */
int cc_normalize_rows_f32(const float* in, float* out, int32_t R, int32_t E, void* stream) {
    if (!in || !out || R <= 0 || E <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, R, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_scaled_dot_nt_f32(const float* a, const float* b, int32_t Bt, int32_t Bv, int32_t E, float mult, float* logits,
                         int32_t ldl, void* stream) {
    if (!a || !b || !logits || Bt <= 0 || Bv <= 0 || E <= 0 || (E & 3) || ldl < Bv) return CC_ERR_INVALID;
    dim3 grid((Bv + ST - 1) / ST, (Bt + ST - 1) / ST);
    hipLaunchKernelGGL(dot_nt_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a, b, logits, Bt, Bv, E, ldl,
                       mult);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_loose_similarity_grouped_f32(const float* text, const float* visual, const int64_t* video_mask, int32_t group,
                                    int64_t vis_group_stride, int64_t mask_group_stride, int64_t mask_row_stride,
                                    int64_t mask_col_stride, int32_t Bt, int32_t Bv, int32_t Tn, int32_t E,
                                    float logit_scale, float* logits, int32_t ldl, float* pooled_out, void* ws,
                                    size_t ws_bytes, void* stream) {
    if (!text || !visual || !video_mask || !logits || group <= 0) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_similarity_workspace_bytes(Bt, Bv, E)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const VidAddr ad{group, vis_group_stride, mask_group_stride, mask_row_stride, mask_col_stride};
    if ((long)Bt * Bv <= 4096 && E <= 1024 && Bt > 0 && Bv > 0) {      // one launch for a step's own logits
        hipLaunchKernelGGL(loose_similarity_small_kernel, dim3(Bv), dim3(256), 0, st, text, visual,
                           reinterpret_cast<const long long*>(video_mask), ad, logits, ldl, pooled_out, Bt, Bv, Tn, E,
                           expf(logit_scale));
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    float* tn = static_cast<float*>(ws);
    float* vp = pooled_out ? pooled_out
                           : reinterpret_cast<float*>(static_cast<char*>(ws) + cc_align_up((size_t)Bt * E * 4, 256));
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((Bt + 3) / 4), dim3(256), 0, st, text, tn, Bt, E);
    CC_LAUNCH_CHECK();
    int rc = video_pool_launch(visual, video_mask, ad, Bv, Tn, E, vp, stream);
    if (rc) return rc;
    return cc_scaled_dot_nt_f32(tn, vp, Bt, Bv, E, expf(logit_scale), logits, ldl, stream);
}

int cc_loose_similarity_strided_f32(const float* text, const float* visual, const int64_t* video_mask,
                                    int64_t mask_row_stride, int64_t mask_col_stride, int32_t Bt, int32_t Bv, int32_t Tn,
                                    int32_t E, float logit_scale, float* logits, int32_t ldl, float* pooled_out, void* ws,
                                    size_t ws_bytes, void* stream) {
    return cc_loose_similarity_grouped_f32(text, visual, video_mask, Bv > 0 ? Bv : 1, 0, 0, mask_row_stride,
                                           mask_col_stride, Bt, Bv, Tn, E, logit_scale, logits, ldl, pooled_out, ws,
                                           ws_bytes, stream);
}

int cc_loose_similarity_f32(const float* text, const float* visual, const int64_t* video_mask, int32_t Bt, int32_t Bv,
                            int32_t Tn, int32_t E, float logit_scale, float* logits, int32_t ldl, float* pooled_out,
                            void* ws, size_t ws_bytes, void* stream) {
    return cc_loose_similarity_strided_f32(text, visual, video_mask, Tn, 1, Bt, Bv, Tn, E, logit_scale, logits, ldl,
                                           pooled_out, ws, ws_bytes, stream);
}

}  // extern "C"

// ============================================================================ N1: retrieval ranks
// compute_metrics (utils/metrics.py:11-26) sorts every row of -sim and looks the diagonal value up.
// Equivalent without a sort: for row i with ground-truth column g = diag_offset + i,
//   c_gt = #{j : sim[i,j] >  sim[i,g]},  c_eq = #{j : sim[i,j] == sim[i,g]}  (>= 1)
// and the reference's `ind` is the concatenation over rows of range(c_gt, c_gt + c_eq) (np.where returns every
// position of the sorted row that equals the diagonal value, so ties contribute several entries).
// One wave per row; element (i, j) lives at sim + i*row_stride + j*col_stride (col_stride != 1 ranks sim^T).
// gt_cols != nullptr: the ground-truth column of row i is gt_cols[i] (multi-sentence retrieval: several text rows
// share one video) and a third count is written: #{j < g : sim[i,j] == sim[i,g]}, the position of g among its ties
// under a stable descending sort.
__global__ __launch_bounds__(256) void rank_counts_kernel(const float* __restrict__ sim, int rows, int cols,
                                                          int64_t row_stride, int64_t col_stride, int diag_offset,
                                                          const int* __restrict__ gt_cols, int* __restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const float* row = sim + (int64_t)i * row_stride;
    const int g = gt_cols ? gt_cols[i] : diag_offset + i;
    const float d = row[(int64_t)g * col_stride];
    int gt = 0, eq = 0, before = 0;
    for (int j = lane; j < cols; j += 64) {
        const float v = row[(int64_t)j * col_stride];
        gt += (v > d) ? 1 : 0;
        eq += (v == d) ? 1 : 0;
        before += (v == d && j < g) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        gt += __shfl_xor(gt, o, CC_WAVE);
        eq += __shfl_xor(eq, o, CC_WAVE);
        before += __shfl_xor(before, o, CC_WAVE);
    }
    if (lane == 0) {
        if (gt_cols) {
            counts[3 * i] = gt;
            counts[3 * i + 1] = eq;
            counts[3 * i + 2] = before;
        } else {
            counts[2 * i] = gt;
            counts[2 * i + 1] = eq;
        }
    }
}

extern "C" int cc_rank_counts_cols_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride,
                                       int64_t col_stride, const int32_t* gt_cols, int32_t* counts3, void* stream) {
    if (!sim || !counts3 || !gt_cols || rows <= 0 || cols <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(rank_counts_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), sim, rows,
                       cols, row_stride, col_stride, 0, gt_cols, counts3);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

extern "C" int cc_rank_counts_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                                  int32_t diag_offset, int32_t* counts, void* stream) {
    if (!sim || !counts || rows <= 0 || cols <= 0 || diag_offset < 0 || diag_offset + rows > cols) return CC_ERR_INVALID;
    hipLaunchKernelGGL(rank_counts_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), sim, rows,
                       cols, row_stride, col_stride, diag_offset, (const int*)nullptr, counts);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// ============================================================================ N4 (forward part): contrastive loss
// CrossEn.forward (modules/losses.py:8-18): logpt = log_softmax(sim, -1); loss = mean(-diag(logpt)), and the symmetric
// form CLIP4Clip.forward builds from it (modules/clip4clip.py:250-253): (CrossEn(sim) + CrossEn(sim^T)) / 2.
// One wave per row (row i of sim, or column i through the strides): nce_i = log(sum_j exp(x_j - max)) + max - x_i.
__global__ __launch_bounds__(256) void cross_entropy_rows_kernel(const float* __restrict__ sim, int n, int64_t rs,
                                                                 int64_t cs, float* __restrict__ nce) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float* row = sim + (int64_t)i * rs;
    float mx = -3.0e38f;
    for (int j = lane; j < n; j += 64) mx = fmaxf(mx, row[(int64_t)j * cs]);
    mx = cc_wave_max(mx);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += expf(row[(int64_t)j * cs] - mx);
    s = cc_wave_sum(s);
    if (lane == 0) nce[i] = (logf(s) + mx) - row[(int64_t)i * cs];
}

// out[0] = mean(nce_rows), out[1] = mean(nce_cols), out[2] = (out[0] + out[1]) / 2; one workgroup, fixed order
__global__ __launch_bounds__(256) void contrastive_mean_kernel(const float* __restrict__ nce, int n, float* __restrict__ out) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a += nce[i]; b += nce[n + i]; }
    a = cc_wave_sum(a);
    b = cc_wave_sum(b);
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float l1 = ((red[0][0] + red[0][1]) + (red[0][2] + red[0][3])) / (float)n;
        const float l2 = ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3])) / (float)n;
        out[0] = l1;
        out[1] = l2;
        out[2] = (l1 + l2) / 2.0f;
    }
}

extern "C" int cc_contrastive_loss_f32(const float* sim, int32_t n, int64_t row_stride, int64_t col_stride, float* loss3,
                                       void* ws, size_t ws_bytes, void* stream) {
    if (!sim || !loss3 || n <= 0) return CC_ERR_INVALID;
    if (!ws || ws_bytes < (size_t)n * 2 * sizeof(float)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* nce = static_cast<float*>(ws);
    hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, sim, n, row_stride, col_stride, nce);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, sim, n, col_stride, row_stride, nce + n);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(contrastive_mean_kernel, dim3(1), dim3(256), 0, st, nce, n, loss3);
    CC_LAUNCH_CHECK();
    return CC_OK;
}
