// Retrieval tail on gfx950: per-frame L2 normalise -> masked mean over the
// temporal segments -> L2 normalise (CLIP4Clip._mean_pooling_for_similarity_visual + _loose_similarity,
// modules/clip4clip.py:305-316,357-366), and the text x video cosine-logit matrix as ONE exact-fp32
// MFMA "NT" GEMM per row block instead of the reference's batch_size_val^2 tiny matmuls with a
// device->host copy each (main.py:502-534).
#include "cc_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows [R, E] -> rows / |row|   (one wave per row)
// Split form of a normalised value for the fp16 matrix cores: hi = fp16(2^10 x), lo = fp16(2^10 x - hi), 2^10 x = hi + lo
// to 22 bits.  For the components that carry a unit row (|x| >= 2^-14: hi normal) lo is at most 2^-11 |hi| and becomes a
// SUBNORMAL fp16 number for |x| below ~2^-3: the scheme relies on v_mfma_f32_16x16x32_f16 multiplying fp16 denormals
// exactly (it does on gfx950, whatever the wave's denormal mode; tests/test_r4_gpu.py::test_similarity_tiny_components
// holds a row dominated by tiny components against float64 - flushed lo parts would show as 2^-11 relative errors).  A row is written as THREE planes side by side, [3E] halfs per row:
//   text  side (A'):  [ hi | hi | lo ]        video side (B'):  [ hi | lo | hi ]
// so that  A' . B'^T = hi.hi + hi.lo + lo.hi  is ONE fp16 GEMM with K = 3E on the main GEMM pipeline (gemm.hip), scaled
// back by the exact factor 2^-20 in its epilogue; the dropped lo.lo term is 2^-22 relative (fp32 rounding level).
struct SplitOut {
    _Float16* hi;       // row-major [R, 3E] or nullptr
    int E;
    int b_side;         // 0: hi | hi | lo,  1: hi | lo | hi
};
__device__ __forceinline__ void split_put(const SplitOut& so, int64_t row, int e, float x) {
    const float sx = x * 1024.0f;
    const _Float16 h = (_Float16)sx;
    const _Float16 l = (_Float16)(sx - (float)h);
    _Float16* p = so.hi + row * (3 * (int64_t)so.E) + e;
    p[0] = h;
    p[so.E] = so.b_side ? l : h;
    p[2 * so.E] = so.b_side ? h : l;
}

typedef _Float16 sh8 __attribute__((ext_vector_type(8)));

// one wave: row r of `in` -> row / |row| (fp32 `out`, optional) and / or its split planes.  The norm is summed exactly as
// before (lane l takes elements l, l + 64, ... in order, then the wave sum), so every path that normalises a row yields
// the same bits; the planes are then written from a second, lane-contiguous read of the row (an L1 hit) as 16-byte
// stores - 2-byte stores strided over the wave were 24 partial-line store instructions per 512-wide row.
__device__ __forceinline__ void normalize_row_wave(const float* __restrict__ in, float* __restrict__ out, const SplitOut& so,
                                                   int r, int E, int lane) {
    const float* src = in + (int64_t)r * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
    const float nrm = sqrtf(cc_wave_sum_fast(s));
    if (so.hi && (E & 7) == 0) {
        _Float16* p = so.hi + (int64_t)r * (3 * (int64_t)E);
        for (int e0 = lane * 8; e0 < E; e0 += 512) {
            const float4 a = *reinterpret_cast<const float4*>(src + e0), b = *reinterpret_cast<const float4*>(src + e0 + 4);
            const float x[8] = {a.x / nrm, a.y / nrm, a.z / nrm, a.w / nrm, b.x / nrm, b.y / nrm, b.z / nrm, b.w / nrm};
            sh8 H, Lo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float sx = x[u] * 1024.0f;
                H[u] = (_Float16)sx;
                Lo[u] = (_Float16)(sx - (float)H[u]);
            }
            *reinterpret_cast<sh8*>(p + e0) = H;
            *reinterpret_cast<sh8*>(p + E + e0) = so.b_side ? Lo : H;
            *reinterpret_cast<sh8*>(p + 2 * E + e0) = so.b_side ? H : Lo;
            if (out) {
                *reinterpret_cast<float4*>(out + (int64_t)r * E + e0) = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(out + (int64_t)r * E + e0 + 4) = make_float4(x[4], x[5], x[6], x[7]);
            }
        }
        return;
    }
    for (int e = lane; e < E; e += 64) {
        const float v = src[e] / nrm;
        if (out) out[(int64_t)r * E + e] = v;
        if (so.hi) split_put(so, r, e, v);
    }
}

__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             SplitOut so, int R, int E) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    normalize_row_wave(in, out, so, r, E, threadIdx.x & 63);
}

// rows that are normalised already -> split planes (the pre-pooled branch, where the video side arrives normalised)
__global__ __launch_bounds__(256) void split_rows_kernel(const float* __restrict__ in, SplitOut so, int64_t total) {
    if ((so.E & 7) == 0) {                                     // 8 consecutive elements per thread: 16-byte plane stores
        const int E = so.E;
        for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < total; i += (int64_t)gridDim.x * 2048) {
            const float4 a = *reinterpret_cast<const float4*>(in + i), b = *reinterpret_cast<const float4*>(in + i + 4);
            const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            sh8 H, Lo;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float sx = x[u] * 1024.0f;
                H[u] = (_Float16)sx;
                Lo[u] = (_Float16)(sx - (float)H[u]);
            }
            const int64_t r = i / E;
            _Float16* p = so.hi + r * (3 * (int64_t)E) + (i - r * E);
            *reinterpret_cast<sh8*>(p) = H;
            *reinterpret_cast<sh8*>(p + E) = so.b_side ? Lo : H;
            *reinterpret_cast<sh8*>(p + 2 * E) = so.b_side ? H : Lo;
        }
        return;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        split_put(so, i / so.E, (int)(i % so.E), in[i]);
}

// visual [Bv, Tn, E], mask [Bv, Tn] int64 -> pooled [Bv, E]   (one wave per video)
// Videos may come in groups of `vg` (the packed all-gather buffer: one record per rank): video v = (group, local) lives at
// visual + group*vgs + local*Tn*E, its mask row at mask + group*mgs + local*mrs (+ t*mcs).
struct VidAddr {
    int vg;
    int64_t vgs, mgs, mrs, mcs;
};
__device__ __forceinline__ void video_pool_wave(const float* __restrict__ visual, const long long* __restrict__ mask,
                                                const VidAddr& ad, float* __restrict__ pooled, const SplitOut& so, int v,
                                                int Tn, int E, int lane) {
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
            if (so.hi) split_put(so, v, e, pv);
        }
    }
}

__global__ __launch_bounds__(256) void video_pool_kernel(const float* __restrict__ visual,
                                                         const long long* __restrict__ mask, VidAddr ad,
                                                         float* __restrict__ pooled, SplitOut so, int Bv, int Tn, int E) {
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= Bv) return;
    video_pool_wave(visual, mask, ad, pooled, so, v, Tn, E, threadIdx.x & 63);
}

// Both producers of the similarity GEMM's operands in ONE launch: workgroups [0, text_blocks) normalise 4 text rows each
// (normalize_rows_kernel), the rest pool 4 videos each (video_pool_kernel) - two latency-bound launches become one.
__global__ __launch_bounds__(256) void sim_prepare_kernel(const float* __restrict__ text, SplitOut ta, int Bt, int text_blocks,
                                                          const float* __restrict__ visual, const long long* __restrict__ mask,
                                                          VidAddr ad, float* __restrict__ pooled, SplitOut vb, int Bv, int Tn,
                                                          int E) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if ((int)blockIdx.x < text_blocks) {
        const int r = blockIdx.x * 4 + w;
        if (r < Bt) normalize_row_wave(text, nullptr, ta, r, E, lane);
    } else {
        const int v = ((int)blockIdx.x - text_blocks) * 4 + w;
        if (v < Bv) video_pool_wave(visual, mask, ad, pooled, vb, v, Tn, E, lane);
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

// The NT product itself: cc_gemm_dispatch (gemm.hip) on the concatenated planes - 256 / 128-row tiles staged by LDS-DMA,
// XOR-swizzled LDS, the half-shifted main loop - with the fp32 epilogue scaling by mult * 2^-20 and dropping the columns
// beyond Bv (the video rows are padded to the tile width inside the workspace; what the padding rows hold is never stored).
// Round 2 ran a 64x64-tile kernel of its own here (12 MFMAs per 8 fragment reads, 2,512 workgroups: 67 us for 10k x 1k,
// bound by neither the matrix pipe nor memory).
// Tile for the [Bt, Bv] product: large problems run ONE round of 8-wave tiles, 256x256 or 256x192 - whichever covers
// the matrix in a round with less work per tile (10k x 1k: 160 tiles of 256x256 leave 96 CUs idle, 240 tiles of 256x192
// do 3/4 of the work each: 43 -> 33 us); everything else is left to the dispatcher's own choice.
static inline int sim_tile(int Bt, int Bv, int* bn_out) {
    const long mt = (Bt + 255) / 256, t256 = mt * ((Bv + 255) / 256), t192 = mt * ((Bv + 191) / 192);
    *bn_out = 256;
    if (t256 < 128) return 0;
    const long c256 = (t256 + 255) / 256 * 4, c192 = (t192 + 255) / 256 * 3;        // rounds x relative tile work
    if (c192 < c256) { *bn_out = 192; return 7; }
    return 5;
}
// rows of the video-side planes inside the workspace: room for either tile width
static inline size_t sim_rows_pad(int Bv) {
    const size_t p256 = ((size_t)Bv + 255) / 256 * 256, p192 = ((size_t)Bv + 191) / 192 * 192;
    return p256 > p192 ? p256 : p192;
}

extern "C" {

size_t cc_similarity_workspace_bytes(int32_t Bt, int32_t Bv, int32_t E) {
    if (Bt <= 0 || Bv <= 0 || E <= 0) return 0;
    return cc_align_up((size_t)Bt * E * 6, 256) + cc_align_up(sim_rows_pad(Bv) * E * 6, 256);
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
    return video_pool_launch(visual, video_mask, ad, Bv, Tn, E, pooled, SplitOut{nullptr, 0, 0}, stream);
}

/* rows [R, E] -> rows / |row| (the text half of _loose_similarity, modules/clip4clip.py:361-362) */
int cc_normalize_rows_f32(const float* in, float* out, int32_t R, int32_t E, void* stream) {
    if (!in || !out || R <= 0 || E <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       SplitOut{nullptr, 0, 0}, R, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// concatenated planes of the text rows ([Bt, 3E]) and the video rows ([pad(Bv), 3E]) inside the similarity workspace
static void sim_planes(void* ws, int Bt, int Bv, int E, SplitOut& ta, SplitOut& vb) {
    _Float16* t = static_cast<_Float16*>(ws);
    ta = SplitOut{t, E, 0};
    _Float16* v = reinterpret_cast<_Float16*>(static_cast<char*>(ws) + cc_align_up((size_t)Bt * E * 6, 256));
    vb = SplitOut{v, E, 1};
}

// products: how many of the three fp16 products of a multiply-add the GEMM issues - the planes are laid out so that the first
// `products` planes of both sides are the right ones: 3 = hi.hi + hi.lo + lo.hi (the operands to 22 bits: 2e-6 on a cosine of
// unit rows, the reference's fp32 product to its own rounding - what the metric fixtures are pinned with), 2 = hi.(hi + lo) =
// fp16(text) x the video operand to 22 bits (the text side rounded to fp16: ~1e-5 rms on a cosine), 1 = hi.hi (both sides
// fp16: ~1.5e-5 rms) - all inside the 1e-3 the contract asks of similarities, at 2/3 and 1/3 of the matrix-core work.
static int dot_planes_launch(const SplitOut& ta, const SplitOut& vb, int Bt, int Bv, int E, float mult, float* logits,
                             int ldl, hipStream_t st, bool zero_pad = true, int products = 3) {
    if (products < 1 || products > 3) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = ta.hi;
    g.W = vb.hi;
    g.C = logits;
    int bn = 256;
    const int tile = sim_tile(Bt, Bv, &bn);
    g.M = Bt; g.N = (Bv + bn - 1) / bn * bn; g.K = products * E; g.ldc = ldl;
    g.lda = g.ldw = 3 * E;
    g.n_valid = Bv;
    // the tiles read video rows up to the padded count: their products are dropped (n_valid), but they are read - zeros
    // instead of whatever the workspace held (uninitialised reads under sanitizers, NaN patterns through the matrix cores)
    if (zero_pad && g.N > Bv && hipMemsetAsync(vb.hi + (size_t)Bv * 3 * E, 0, (size_t)(g.N - Bv) * 3 * E * sizeof(_Float16), st) != hipSuccess)
        return CC_ERR_HIP;
    g.out_scale = mult * 9.5367431640625e-07f;                // 2^-20: undo the two 2^10 operand scalings (exact)
    return cc_gemm_dispatch(g, EPI_F32, tile, st);
}

int cc_scaled_dot_nt_f32(const float* a, const float* b, int32_t Bt, int32_t Bv, int32_t E, float mult, float* logits,
                         int32_t ldl, void* ws, size_t ws_bytes, void* stream) {
    if (!a || !b || !logits || Bt <= 0 || Bv <= 0 || E <= 0 || (E & 63) || ldl < Bv) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_similarity_workspace_bytes(Bt, Bv, E)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    SplitOut ta, vb;
    sim_planes(ws, Bt, Bv, E, ta, vb);
    const int64_t na = (int64_t)Bt * E, nb = (int64_t)Bv * E;
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((na + 2047) / 2048 < 4096 ? (na + 2047) / 2048 : 4096)), dim3(256), 0, st, a, ta, na);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(split_rows_kernel, dim3((unsigned)((nb + 2047) / 2048 < 4096 ? (nb + 2047) / 2048 : 4096)), dim3(256), 0, st, b, vb, nb);
    CC_LAUNCH_CHECK();
    return dot_planes_launch(ta, vb, Bt, Bv, E, mult, logits, ldl, st);
}

/* ---- the similarity operands as the producers' by-product (eval: S3).  The text / video features of a batch are written
 * as split-fp16 planes when the batch is encoded; the Nt x Nv matrix at the end of the epoch is then the GEMM alone.
 * A plane row is 3E halfs: text side [hi | hi | lo], video side [hi | lo | hi] (see the top of this file). */
size_t cc_similarity_plane_row_bytes(int32_t E) { return E > 0 ? (size_t)E * 6 : 0; }
/* rows of a video-side plane buffer the GEMM reads for Bv videos (whole tiles): allocate and ZERO this many */
int32_t cc_similarity_padded_rows(int32_t Bv) { return Bv > 0 ? (int32_t)sim_rows_pad(Bv) : 0; }

int cc_normalize_rows_planes_f32(const float* in, float* out, void* planes, int32_t video_side, int32_t R, int32_t E,
                                 void* stream) {
    if (!in || !planes || R <= 0 || E <= 0 || (E & 63)) return CC_ERR_INVALID;
    hipLaunchKernelGGL(normalize_rows_kernel, dim3((R + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), in, out,
                       SplitOut{static_cast<_Float16*>(planes), E, video_side ? 1 : 0}, R, E);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_video_pool_normalize_planes_f32(const float* visual, const int64_t* video_mask, int32_t Bv, int32_t Tn, int32_t E,
                                       float* pooled, void* planes, void* stream) {
    if (!planes || (E & 63)) return CC_ERR_INVALID;
    const VidAddr ad{Bv > 0 ? Bv : 1, 0, 0, Tn, 1};
    return video_pool_launch(visual, video_mask, ad, Bv, Tn, E, pooled, SplitOut{static_cast<_Float16*>(planes), E, 1}, stream);
}

int cc_scaled_dot_planes_f32(const void* text_planes, const void* video_planes, int32_t Bt, int32_t Bv,
                             int32_t video_rows, int32_t E, float mult, float* logits, int32_t ldl, void* stream) {
    return cc_scaled_dot_planes_products_f32(text_planes, video_planes, Bt, Bv, video_rows, E, mult, 3, logits, ldl, stream);
}

int cc_scaled_dot_planes_products_f32(const void* text_planes, const void* video_planes, int32_t Bt, int32_t Bv,
                                      int32_t video_rows, int32_t E, float mult, int32_t products, float* logits, int32_t ldl,
                                      void* stream) {
    if (!text_planes || !video_planes || !logits || Bt <= 0 || Bv <= 0 || E <= 0 || (E & 63) || ldl < Bv) return CC_ERR_INVALID;
    if (products < 1 || products > 3) return CC_ERR_INVALID;
    int bn = 256;
    (void)sim_tile(Bt, Bv, &bn);
    if (video_rows < (Bv + bn - 1) / bn * bn) return CC_ERR_WORKSPACE;       // the tiles read whole multiples of their width
    const SplitOut ta{const_cast<_Float16*>(static_cast<const _Float16*>(text_planes)), E, 0};
    const SplitOut vb{const_cast<_Float16*>(static_cast<const _Float16*>(video_planes)), E, 1};
    return dot_planes_launch(ta, vb, Bt, Bv, E, mult, logits, ldl, static_cast<hipStream_t>(stream), false, products);
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
    if (E > 1024 || Tn <= 0) return E > 1024 ? CC_ERR_UNSUPPORTED : CC_ERR_INVALID;
    const int tb = (Bt + 3) / 4;
    hipLaunchKernelGGL(sim_prepare_kernel, dim3(tb + (Bv + 3) / 4), dim3(256), 0, st, text, ta, Bt, tb, visual,
                       reinterpret_cast<const long long*>(video_mask), ad, pooled_out, vb, Bv, Tn, E);
    CC_LAUNCH_CHECK();
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

// ---------------------------------------------------------------------------- multi-sentence video -> text maxima
// tensor_video_to_text_sim (utils/metrics.py:68-76): for every (group of sentences, video) the best similarity of the
// group's sentences; NaN entries (fully masked clips) count as -inf.  out [n_groups, cols] must hold -inf beforehand
// (group_max_fill_kernel); a thread walks one column over a chunk of rows, keeps the running maximum while the group id
// repeats (sentences of one video are neighbours) and publishes it with an ordered atomic on a group change - so rows of
// one group may also be spread over chunks, or over the row blocks of several ranks.
__global__ void group_max_fill_kernel(float* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = -INFINITY;
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v) {      // (finite or -inf values; initial -inf)
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
constexpr int GMAX_ROWS = 64;       // rows per chunk
__global__ __launch_bounds__(256) void group_max_rows_kernel(const float* __restrict__ sim, int rows, int cols,
                                                             int64_t row_stride, const int* __restrict__ group,
                                                             int n_groups, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r0 = blockIdx.y * GMAX_ROWS, r1 = min(rows, r0 + GMAX_ROWS);
    if (c >= cols) return;
    int gcur = -1;
    float m = -INFINITY;
    for (int r = r0; r < r1; ++r) {
        const int g = group[r];
        if (g != gcur) {
            if (gcur >= 0 && gcur < n_groups) atomic_max_f32(out + (int64_t)gcur * cols + c, m);
            gcur = g;
            m = -INFINITY;
        }
        const float v = sim[(int64_t)r * row_stride + c];
        m = (v == v && v > m) ? v : m;
    }
    if (gcur >= 0 && gcur < n_groups) atomic_max_f32(out + (int64_t)gcur * cols + c, m);
}

extern "C" int cc_group_max_rows_f32(const float* sim, int32_t rows, int32_t cols, int64_t row_stride, const int32_t* group,
                                     int32_t n_groups, float* out, void* stream) {
    if (!out || n_groups <= 0 || cols <= 0 || rows < 0 || (rows > 0 && (!sim || !group))) return CC_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n = (int64_t)n_groups * cols;
    hipLaunchKernelGGL(group_max_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n);
    if (rows > 0)
        hipLaunchKernelGGL(group_max_rows_kernel, dim3((cols + 255) / 256, (rows + GMAX_ROWS - 1) / GMAX_ROWS), dim3(256), 0, st,
                           sim, rows, cols, row_stride, group, n_groups, out);
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

// ============================================================================ N4: the training tail with its gradient
// loss = (CrossEn(S) + CrossEn(S^T)) / 2,  S = exp(logit_scale) * t_hat p_hat^T  (modules/clip4clip.py:245-262,357-366,
// modules/losses.py:8-18), t_hat = t / |t|, p = sum_f m_f (v_f / |v_f|) / max(sum_f m_f, 1), p_hat = p / |p|.
// What torch.autograd derives for it (g = the incoming gradient of the loss, c = exp(logit_scale), n = batch):
//   G = dL/dS = g / (2n) * (softmax_rows(S) + softmax_cols(S) - 2 I)          d logit_scale = sum_ij G_ij S_ij
//   d t_hat = c G p_hat              d t = (d t_hat - t_hat (t_hat . d t_hat)) / |t|
//   d p_hat = c G^T t_hat            d p = (d p_hat - p_hat (p_hat . d p_hat)) / |p|
//   d vhat_f = d p * m_f / den       d v_f = (d vhat_f - vhat_f (vhat_f . d vhat_f)) / |v_f|
// Everything fp32 in fixed summation orders (one wave per row, lane-strided partial sums + the shuffle tree): no atomics,
// the same bits on every run.  n is a training batch (<= a few hundred): the n x n x E products are one wave per output row.
namespace {

struct CtrWs {
    float* that;   // [n, E]
    float* phat;   // [n, E]
    float* tn;     // [n]  |t|
    float* pn;     // [n]  |p|
    float* den;    // [n]  max(sum mask, 1)
    float* vn;     // [n, Tn]  |v_f|
    float* S;      // [n, n]
    float* nce;    // [2n]  rows | cols
    float* G;      // [n, n]
    float* dls;    // [n]  per-row partial of d logit_scale
    size_t total;
};

CtrWs ctr_carve(void* ws, int n, int Tn, int E) {
    CtrWs c{};
    size_t off = 0;
    auto take = [&](size_t floats) {
        float* p = ws ? reinterpret_cast<float*>(static_cast<char*>(ws) + off) : nullptr;
        off += cc_align_up(floats * sizeof(float), 256);
        return p;
    };
    c.that = take((size_t)n * E); c.phat = take((size_t)n * E);
    c.tn = take(n); c.pn = take(n); c.den = take(n); c.vn = take((size_t)n * Tn);
    c.S = take((size_t)n * n); c.nce = take(2 * (size_t)n); c.G = take((size_t)n * n); c.dls = take(n);
    c.total = off;
    return c;
}

// one wave per row: [0, n) text rows, [n, 2n) videos
__global__ __launch_bounds__(256) void ctr_prepare_kernel(const float* __restrict__ text, const float* __restrict__ visual,
                                                          const long long* __restrict__ mask, int64_t mrs, int64_t mcs, int n,
                                                          int Tn, int E, CtrWs w) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= 2 * n) return;
    if (r < n) {
        const float* src = text + (int64_t)r * E;
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
        const float nrm = sqrtf(cc_wave_sum(s));
        for (int e = lane; e < E; e += 64) w.that[(int64_t)r * E + e] = src[e] / nrm;
        if (lane == 0) w.tn[r] = nrm;
        return;
    }
    const int v = r - n;
    float cnt = 0.f;
    for (int t = 0; t < Tn; ++t) cnt += (float)mask[(int64_t)v * mrs + (int64_t)t * mcs];
    if (cnt == 0.f) cnt = 1.f;                    // "avoid zero divide", clip4clip.py:313
    float sp = 0.f;
    for (int e = lane; e < E; e += 64) w.phat[(int64_t)v * E + e] = 0.f;
    for (int t = 0; t < Tn; ++t) {
        const float* src = visual + ((int64_t)v * Tn + t) * E;
        float s = 0.f;
        for (int e = lane; e < E; e += 64) s = fmaf(src[e], src[e], s);
        const float nrm = sqrtf(cc_wave_sum(s));
        const float mk = (float)mask[(int64_t)v * mrs + (int64_t)t * mcs];
        for (int e = lane; e < E; e += 64) w.phat[(int64_t)v * E + e] += (src[e] / nrm) * mk;   // (same lane re-reads its own element)
        if (lane == 0) w.vn[(int64_t)v * Tn + t] = nrm;
    }
    for (int e = lane; e < E; e += 64) {
        const float p = w.phat[(int64_t)v * E + e] / cnt;
        w.phat[(int64_t)v * E + e] = p;
        sp = fmaf(p, p, sp);
    }
    const float pn = sqrtf(cc_wave_sum(sp));
    for (int e = lane; e < E; e += 64) w.phat[(int64_t)v * E + e] /= pn;
    if (lane == 0) { w.pn[v] = pn; w.den[v] = cnt; }
}

// S[i][j] = c * that_i . phat_j: one wave per (i, j)
__global__ __launch_bounds__(256) void ctr_sim_kernel(CtrWs w, int n, int E, float c, const float* __restrict__ ls_dev) {
    if (ls_dev) c = expf(*ls_dev);                               // (training: the parameter itself, no host read of it)
    const int lane = threadIdx.x & 63;
    const int64_t idx = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= (int64_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    const float* a = w.that + (int64_t)i * E;
    const float* b = w.phat + (int64_t)j * E;
    float s = 0.f;
    for (int e = lane; e < E; e += 64) s = fmaf(a[e], b[e], s);
    s = cc_wave_sum(s);
    if (lane == 0) w.S[idx] = c * s;
}

// G_ij = g/(2n) (exp(S_ij - lse_row_i) + exp(S_ij - lse_col_j) - 2 [i == j]); row partial of sum_ij G_ij S_ij.  One wave per row.
__global__ __launch_bounds__(256) void ctr_grad_sim_kernel(CtrWs w, int n, float g) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const float lse_r = w.nce[i] + w.S[(int64_t)i * n + i];            // nce_i = lse_i - S_ii (cross_entropy_rows_kernel)
    const float k = g / (2.0f * (float)n);
    float acc = 0.f;
    for (int j = lane; j < n; j += 64) {
        const float s = w.S[(int64_t)i * n + j];
        const float lse_c = w.nce[n + j] + w.S[(int64_t)j * n + j];
        const float gij = k * ((expf(s - lse_r) + expf(s - lse_c)) - (i == j ? 2.0f : 0.0f));
        w.G[(int64_t)i * n + j] = gij;
        acc = fmaf(gij, s, acc);
    }
    acc = cc_wave_sum(acc);
    if (lane == 0) w.dls[i] = acc;
}

// one wave per output row: [0, n) d_text rows, [n, 2n) videos (all Tn frames), row 2n: d logit_scale
__global__ __launch_bounds__(256) void ctr_grad_feat_kernel(const float* __restrict__ visual, const long long* __restrict__ mask,
                                                            int64_t mrs, int64_t mcs, int n, int Tn, int E, float c, CtrWs w,
                                                            float* __restrict__ d_text, float* __restrict__ d_visual,
                                                            float* __restrict__ d_ls, const float* __restrict__ ls_dev) {
    if (ls_dev) c = expf(*ls_dev);
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int MAXE = 16;                                           // E <= 1024
    if (r == 2 * n) {
        float s = 0.f;
        for (int i = lane; i < n; i += 64) s += w.dls[i];
        s = cc_wave_sum(s);
        if (lane == 0) *d_ls = s;                                      // dS/d logit_scale = S
        return;
    }
    if (r > 2 * n) return;
    float d[MAXE];
#pragma unroll
    for (int q = 0; q < MAXE; ++q) d[q] = 0.f;
    if (r < n) {                                                       // d that_i = c sum_j G_ij phat_j
        for (int j = 0; j < n; ++j) {
            const float gij = w.G[(int64_t)r * n + j];
#pragma unroll
            for (int q = 0; q < MAXE; ++q) {
                const int e = lane + 64 * q;
                if (e < E) d[q] = fmaf(gij, w.phat[(int64_t)j * E + e], d[q]);
            }
        }
        float dot = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            d[q] *= c;
            if (e < E) dot = fmaf(w.that[(int64_t)r * E + e], d[q], dot);
        }
        dot = cc_wave_sum(dot);
        const float inv = 1.0f / w.tn[r];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) d_text[(int64_t)r * E + e] = (d[q] - w.that[(int64_t)r * E + e] * dot) * inv;
        }
        return;
    }
    const int v = r - n;                                               // d phat_v = c sum_i G_iv that_i
    for (int i = 0; i < n; ++i) {
        const float giv = w.G[(int64_t)i * n + v];
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) d[q] = fmaf(giv, w.that[(int64_t)i * E + e], d[q]);
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = lane + 64 * q;
        d[q] *= c;
        if (e < E) dot = fmaf(w.phat[(int64_t)v * E + e], d[q], dot);
    }
    dot = cc_wave_sum(dot);
    const float invp = 1.0f / w.pn[v], den = w.den[v];
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = lane + 64 * q;
        if (e < E) d[q] = (d[q] - w.phat[(int64_t)v * E + e] * dot) * invp;      // d p
    }
    for (int t = 0; t < Tn; ++t) {
        const float* src = visual + ((int64_t)v * Tn + t) * E;
        const float mk = (float)mask[(int64_t)v * mrs + (int64_t)t * mcs];
        const float nrm = w.vn[(int64_t)v * Tn + t];
        float dv[MAXE], vh[MAXE];
        float dt = 0.f;
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            vh[q] = e < E ? src[e] / nrm : 0.f;
            dv[q] = d[q] * mk / den;                                   // d vhat_f
            dt = fmaf(vh[q], dv[q], dt);
        }
        dt = cc_wave_sum(dt);
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            const int e = lane + 64 * q;
            if (e < E) d_visual[((int64_t)v * Tn + t) * E + e] = (dv[q] - vh[q] * dt) / nrm;
        }
    }
}

}  // namespace

extern "C" size_t cc_contrastive_grad_workspace_bytes(int32_t n, int32_t Tn, int32_t E) {
    if (n <= 0 || Tn <= 0 || E <= 0) return 0;
    return ctr_carve(nullptr, n, Tn, E).total;
}

extern "C" int cc_contrastive_loss_grad_f32(const float* text, const float* visual, const int64_t* video_mask,
                                            int64_t mask_row_stride, int64_t mask_col_stride, int32_t n, int32_t Tn, int32_t E,
                                            float logit_scale, float grad_scale, float* loss3, float* d_text, float* d_visual,
                                            float* d_logit_scale, void* ws, size_t ws_bytes, void* stream) {
    return cc_contrastive_loss_grad_dev_f32(text, visual, video_mask, mask_row_stride, mask_col_stride, n, Tn, E, logit_scale, nullptr,
                                            grad_scale, loss3, d_text, d_visual, d_logit_scale, ws, ws_bytes, stream);
}

/* the same with logit_scale read from device memory when logit_scale_dev != null (the nn.Parameter itself: a training step then
 * neither synchronises with the host for its value nor bakes it into a captured graph) */
extern "C" int cc_contrastive_loss_grad_dev_f32(const float* text, const float* visual, const int64_t* video_mask,
                                                int64_t mask_row_stride, int64_t mask_col_stride, int32_t n, int32_t Tn, int32_t E,
                                                float logit_scale, const float* logit_scale_dev, float grad_scale, float* loss3,
                                                float* d_text, float* d_visual, float* d_logit_scale, void* ws, size_t ws_bytes,
                                                void* stream) {
    if (!text || !visual || !video_mask || !loss3 || !d_text || !d_visual || !d_logit_scale) return CC_ERR_INVALID;
    if (n <= 0 || Tn <= 0 || E <= 0) return CC_ERR_INVALID;
    if (E > 1024) return CC_ERR_UNSUPPORTED;
    const CtrWs w = ctr_carve(ws, n, Tn, E);
    if (!ws || ws_bytes < w.total) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float c = expf(logit_scale);
    const long long* mk = reinterpret_cast<const long long*>(video_mask);
    hipLaunchKernelGGL(ctr_prepare_kernel, dim3((2 * n + 3) / 4), dim3(256), 0, st, text, visual, mk, mask_row_stride,
                       mask_col_stride, n, Tn, E, w);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ctr_sim_kernel, dim3((unsigned)(((int64_t)n * n + 3) / 4)), dim3(256), 0, st, w, n, E, c, logit_scale_dev);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, w.S, n, (int64_t)n, (int64_t)1, w.nce);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, st, w.S, n, (int64_t)1, (int64_t)n, w.nce + n);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(contrastive_mean_kernel, dim3(1), dim3(256), 0, st, w.nce, n, loss3);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ctr_grad_sim_kernel, dim3((n + 3) / 4), dim3(256), 0, st, w, n, grad_scale);
    CC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ctr_grad_feat_kernel, dim3((2 * n + 1 + 3) / 4), dim3(256), 0, st, visual, mk, mask_row_stride,
                       mask_col_stride, n, Tn, E, c, w, d_text, d_visual, d_logit_scale, logit_scale_dev);
    CC_LAUNCH_CHECK();
    return CC_OK;
}
