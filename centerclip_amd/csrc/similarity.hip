// Retrieval tail on gfx950: per-frame L2 normalise -> masked mean over the
// temporal segments -> L2 normalise (CLIP4Clip._mean_pooling_for_similarity_visual + _loose_similarity,
// modules/clip4clip.py:305-316,357-366), and the text x video cosine-logit matrix as ONE exact-fp32
// MFMA "NT" GEMM per row block instead of the reference's batch_size_val^2 tiny matmuls with a
// device->host copy each (main.py:502-534).
#include "cc_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows [R, E] -> rows / |row|   (one wave per row)
// Split form of a normalised value for the fp16 matrix cores (see dot_nt_kernel): hi = fp16(2^10 x), lo = fp16(2^10 x - hi)
struct SplitOut {
    _Float16* hi;       // [R, E] or nullptr
    _Float16* lo;
};
__device__ __forceinline__ void split_put(const SplitOut& so, int64_t idx, float x) {
    const float sx = x * 1024.0f;
    const _Float16 h = (_Float16)sx;
    so.hi[idx] = h;
    so.lo[idx] = (_Float16)(sx - (float)h);
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             SplitOut so, int R, int E) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* src = in + (int64_t)r * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
    const float nrm = sqrtf(cc_wave_sum_fast(s));
    for (int e = lane; e < E; e += 64) {
        const float v = src[e] / nrm;
        if (out) out[(int64_t)r * E + e] = v;
        if (so.hi) split_put(so, (int64_t)r * E + e, v);
    }
}

// rows that are normalised already -> split planes (the pre-pooled branch, where the video side arrives normalised)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, SplitOut so, int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) split_put(so, i, in[i]);
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
                                                         float* __restrict__ pooled, SplitOut so, int Bv, int Tn, int E) {
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
        const float nrm = sqrtf(cc_wave_sum_fast(s));
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
    const float nrm = sqrtf(cc_wave_sum_fast(s));
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = lane + 64 * q;
        if (e < E) {
            const float pv = acc[q] / nrm;
            if (pooled) pooled[(int64_t)v * E + e] = pv;
            if (so.hi) split_put(so, (int64_t)v * E + e, pv);
        }
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
    const int v = blockIdx.x;                     // grid.y = groups of 4 texts: every wave owns ONE text (a workgroup per
    constexpr int MAXE = 16;                      // video made each wave walk Bt / 4 texts in a dependent chain); E <= 1024
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
            const float nrm = sqrtf(cc_wave_sum_fast(s));
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
        const float nrm = sqrtf(cc_wave_sum_fast(s));
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) {
                const float p = acc[q] / nrm;
                vp[e] = p;
                if (pooled_out && blockIdx.y == 0) pooled_out[(int64_t)v * E + e] = p;
            }
        }
    }
    __syncthreads();
    for (int t = (int)blockIdx.y * 4 + wave; t < Bt; t += 4 * (int)gridDim.y) {
        const float* src = text + (int64_t)t * E;
        float x[MAXE];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            x[q] = e < E ? src[e] : 0.f;
            s = fmaf(x[q], x[q], s);
        }
        const float nrm = sqrtf(cc_wave_sum_fast(s));
        float d = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) d = fmaf(x[q] / nrm, vp[e], d);
        }
        d = cc_wave_sum_fast(d);
        if (lane == 0) logits[(int64_t)t * ldl + v] = mult * d;
    }
}

// C[i][j] = mult * sum_k A[i][k] B[j][k]; 64x64 tile per workgroup.
// Arithmetic as in the Gram kernel of cluster.hip: each normalised fp32 value x (|x| <= 1) was split by its producer
// (normalize_rows_kernel / video_pool_kernel / split_rows_kernel) into hi = fp16(2^10 x) and lo = fp16(2^10 x - hi) -
// both in fp16's normal range, 2^10 x = hi + lo to 22 bits - and the dot product is accumulated in fp32 as
// hi.hi + hi.lo + lo.hi on the fp16 matrix cores (16x the rate of the exact-fp32 MFMA the first version used and was
// bound by), then scaled back by the exact factor 2^-20.  The dropped lo.lo term is 2^-22 relative: fp32 rounding level.
// Splitting once per row instead of in every tile keeps the tile's staging a pure copy: the four operand planes go
// HBM -> LDS by LDS-DMA (16 B per lane, chunks XOR-swizzled through the source address as in gemm.hip).
#define ST 64
#define SKK 64
typedef _Float16 sh8 __attribute__((ext_vector_type(8)));
#define SPLANE (ST * SKK)
__device__ __forceinline__ void sim_glds16(const _Float16* g, _Float16* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ __launch_bounds__(256) void dot_nt_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                     const _Float16* __restrict__ Bh, const _Float16* __restrict__ Bl,
                                                     float* __restrict__ C, int M, int N, int K, int ldc, float mult) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][4][SPLANE];      // [buffer][Ah, Al, Bh, Bl]
    const int ti = blockIdx.y, tj = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // staging: LDS chunk idx (16 B) = q*256 + tid -> row idx / 8, position idx % 8; source chunk = position ^ (row & 7)
    const _Float16* src[4][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = q * 256 + tid, r = idx >> 3, c = (idx & 7) ^ (r & 7);
        const int64_t ra = (int64_t)min(ti * ST + r, M - 1) * K + c * 8, rb = (int64_t)min(tj * ST + r, N - 1) * K + c * 8;
        src[0][q] = Ah + ra; src[1][q] = Al + ra; src[2][q] = Bh + rb; src[3][q] = Bl + rb;
    }
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int pl = 0; pl < 4; ++pl)
#pragma unroll
            for (int q = 0; q < 2; ++q) sim_glds16(src[pl][q] + kt * SKK, &lds[buf][pl][(q * 4 + wave) * 512]);
    };
    const int wr = wave >> 1, wc = wave & 1;
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = K / SKK;                                    // K % 64 == 0 (checked by the launcher)
    stage(0, 0);
    __syncthreads();
    const int g = lane >> 4, l15 = lane & 15;
    auto frag = [&](const _Float16* pl, int row, int ks) {     // 8 consecutive k of `row` at k = ks*32 + g*8
        return *reinterpret_cast<const sh8*>(pl + row * SKK + ((((ks << 2) | g) ^ (row & 7)) << 3));
    };
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const int ar = wr * 32 + l15, br = wc * 32 + l15;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const sh8 a0h = frag(lds[buf][0], ar, ks), a0l = frag(lds[buf][1], ar, ks);
            const sh8 a1h = frag(lds[buf][0], ar + 16, ks), a1l = frag(lds[buf][1], ar + 16, ks);
            const sh8 b0h = frag(lds[buf][2], br, ks), b0l = frag(lds[buf][3], br, ks);
            const sh8 b1h = frag(lds[buf][2], br + 16, ks), b1l = frag(lds[buf][3], br + 16, ks);
            // B fragment as the first operand: a lane then owns 4 consecutive j of one i (16-byte stores)
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0h, a0h, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0h, a0l, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0l, a0h, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1h, a0h, acc[0][1], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1h, a0l, acc[0][1], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1l, a0h, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0h, a1h, acc[1][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0h, a1l, acc[1][0], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0l, a1h, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1h, a1h, acc[1][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1h, a1l, acc[1][1], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1l, a1h, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    const float sc = mult * 9.5367431640625e-07f;              // 2^-20: undo the two 2^10 operand scalings (exact)
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
                *reinterpret_cast<float4*>(dst) = make_float4(sc * v[0], sc * v[1], sc * v[2], sc * v[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (j + e < N) dst[e] = sc * v[e];
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
                             int32_t Tn, int32_t E, float* pooled, SplitOut so, void* stream) {
    if (!visual || !video_mask || (!pooled && !so.hi) || Bv <= 0 || Tn <= 0 || E <= 0 || ad.vg <= 0) return CC_ERR_INVALID;
    if (E > 1024) return CC_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(video_pool_kernel, dim3((Bv + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), visual,
                       reinterpret_cast<const long long*>(video_mask), ad, pooled, so, Bv, Tn, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_video_pool_normalize_f32(const float* visual, const int64_t* video_mask, int32_t Bv, int32_t Tn, int32_t E,
                                float* pooled, void* stream) {
    const VidAddr ad{Bv > 0 ? Bv : 1, 0, 0, Tn, 1};
    return video_pool_launch(visual, video_mask, ad, Bv, Tn, E, pooled, SplitOut{nullptr, nullptr}, stream);
}

/* rows [R, E] -> rows / |row| (the text half of _loose_similarity, modules/clip4clip.py:361-362) */
int cc_normalize_rows_f32(const float* in, float* out, int32_t R, int32_t E, void* stream) {
    if (!in || !out || R <= 0 || E <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       SplitOut{nullptr, nullptr}, R, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// split planes of the text rows ([Bt,E] hi | lo) and the video rows ([Bv,E] hi | lo) inside the similarity workspace
// (cc_similarity_workspace_bytes = (Bt + Bv) * E floats, each region exactly the size of its two fp16 planes)
static void sim_planes(void* ws, int Bt, int Bv, int E, SplitOut& ta, SplitOut& vb) {
    _Float16* t = static_cast<_Float16*>(ws);
    ta = SplitOut{t, t + (size_t)Bt * E};
    _Float16* v = reinterpret_cast<_Float16*>(static_cast<char*>(ws) + cc_align_up((size_t)Bt * E * 4, 256));
    vb = SplitOut{v, v + (size_t)Bv * E};
}

static int dot_planes_launch(const SplitOut& ta, const SplitOut& vb, int Bt, int Bv, int E, float mult, float* logits,
                             int ldl, hipStream_t st) {
    dim3 grid((Bv + ST - 1) / ST, (Bt + ST - 1) / ST);
    hipLaunchKernelGGL(dot_nt_kernel, grid, dim3(256), 0, st, ta.hi, ta.lo, vb.hi, vb.lo, logits, Bt, Bv, E, ldl, mult);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_scaled_dot_nt_f32(const float* a, const float* b, int32_t Bt, int32_t Bv, int32_t E, float mult, float* logits,
                         int32_t ldl, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !logits || Bt <= 0 || Bv <= 0 || E <= 0 || (E & 63) || ldl < Bv) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_similarity_workspace_bytes(Bt, Bv, E)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SplitOut ta, vb;
    sim_planes(ws, Bt, Bv, E, ta, vb);
    const int64_t na = (int64_t)Bt * E, nb = (int64_t)Bv * E;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((na + 255) / 256 < 4096 ? (na + 255) / 256 : 4096)), dim3(256), 0, st, a, ta, na);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((nb + 255) / 256 < 4096 ? (nb + 255) / 256 : 4096)), dim3(256), 0, st, b, vb, nb);
    CC_LAUNCH_CHECK();
    return dot_planes_launch(ta, vb, Bt, Bv, E, mult, logits, ldl, st);
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
        hipLaunchKernelGGL(loose_similarity_small_kernel, dim3(Bv, (Bt + 3) / 4), dim3(256), 0, st, text, visual,
                           reinterpret_cast<const long long*>(video_mask), ad, logits, ldl, pooled_out, Bt, Bv, Tn, E,
                           expf(logit_scale));
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    if (E & 63) return CC_ERR_UNSUPPORTED;                    // the NT GEMM walks K in steps of 64
    // text rows: normalise -> split planes; videos: pool + normalise -> split planes (+ fp32 pooled_out); then the GEMM
    SplitOut ta, vb;
    sim_planes(ws, Bt, Bv, E, ta, vb);
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((Bt + 3) / 4), dim3(256), 0, st, text, (float*)nullptr, ta, Bt, E);
    CC_LAUNCH_CHECK();
    int rc = video_pool_launch(visual, video_mask, ad, Bv, Tn, E, pooled_out, vb, stream);
    if (rc) return rc;
    return dot_planes_launch(ta, vb, Bt, Bv, E, expf(logit_scale), logits, ldl, st);
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
                                                          const int* __restrict__ gt_cols, int* __restrict__ counts,
                                                          const float* __restrict__ ref_vals) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const float* row = sim + (int64_t)i * row_stride;
    const int g = gt_cols ? gt_cols[i] : diag_offset + i;
    // ref_vals: the value to rank against is handed in (row-sharded matrices: the ground-truth entry of a COLUMN lives
    // in another rank's row block) instead of being read from the row itself
    const float d = ref_vals ? ref_vals[i] : row[(int64_t)g * col_stride];
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
                       cols, row_stride, col_stride, 0, gt_cols, counts3, (const float*)nullptr);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

extern "C" int cc_rank_counts_ref_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                                      const float* ref_vals, int32_t* counts, void* stream) {
    if (!sim || !counts || !ref_vals || rows <= 0 || cols <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(rank_counts_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), sim, rows,
                       cols, row_stride, col_stride, 0, (const int*)nullptr, counts, ref_vals);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

extern "C" int cc_rank_counts_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, int64_t col_stride,
                                  int32_t diag_offset, int32_t* counts, void* stream) {
    if (!sim || !counts || rows <= 0 || cols <= 0 || diag_offset < 0 || diag_offset + rows > cols) return CC_ERR_INVALID;
    hipLaunchKernelGGL(rank_counts_kernel, dim3((rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), sim, rows,
                       cols, row_stride, col_stride, diag_offset, (const int*)nullptr, counts, (const float*)nullptr);
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
    s = cc_wave_sum_fast(s);
    if (lane == 0) nce[i] = (logf(s) + mx) - row[(int64_t)i * cs];
}

// out[0] = mean(nce_rows), out[1] = mean(nce_cols), out[2] = (out[0] + out[1]) / 2; one workgroup, fixed order
__global__ __launch_bounds__(256) void contrastive_mean_kernel(const float* __restrict__ nce, int n, float* __restrict__ out) {
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) { a += nce[i]; b += nce[n + i]; }
    a = cc_wave_sum_fast(a);
    b = cc_wave_sum_fast(b);
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
