// N4, first slice of training: the backward pieces of ONE ResidualAttentionBlock (modules/clip.py:228-253; the reference
// obtains them from torch.autograd inside main.py:321 scaler.scale(loss).backward()).  The contractions (dgrad / wgrad of
// the four Linear layers) run on the forward GEMM kernel with swapped operand roles (centerclip_amd/train.py); this file
// holds what is not a GEMM:
//   layernorm_backward      dx = rstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma;  dgamma = sum dy xhat, dbeta = sum dy
//   quick_gelu_backward     d/dx [x sigmoid(1.702 x)] = s + 1.702 x s (1 - s)
//   attention_backward      dV = P^T dO, dP = dO V^T, dS = P (dP - rowsum(dP P)), dQ = dS K / 8, dK = dS^T Q / 8   (L <= 256)
//   column_sums             bias gradients
//   absmax / cast_scaled / unscale   gradients travel through the fp16 matrix cores with a per-tensor power-of-two scale
//                           chosen on the device (no host synchronisation), removed again from the fp32 product
// fp32 arithmetic throughout; every reduction has a fixed order (no atomics on floats): identical bits on every run.
#include "cc_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int LNB_ROWS = 32;        // rows per workgroup (4 waves x 8 rows)

// The largest magnitude of what a workgroup wrote, into *bits (non-negative floats order as their bit patterns): the next
// cc_cast_transpose_f16 of that tensor (scaled = 2) then needs no pass of its own over it.  One atomic per workgroup; every
// thread of the 256 calls it.
__device__ __forceinline__ void publish_absmax(float m, unsigned* __restrict__ bits) {
    __shared__ float wmax_[4];
    m = cc_wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax_[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        // (a workgroup whose maximum does not exceed the published one skips the atomic - a stale read only costs an atomic that
        //  changes nothing; after the first few workgroups almost all skip, so the grid need not be capped for it)
        const unsigned mine = __float_as_uint(fmaxf(fmaxf(wmax_[0], wmax_[1]), fmaxf(wmax_[2], wmax_[3])));
        if (mine > __atomic_load_n(bits, __ATOMIC_RELAXED)) atomicMax(bits, mine);
    }
}

// One wave per row; lane l owns columns l*4 + t*256 (t < 4: W <= 1024).  Per-workgroup partial sums of dgamma / dbeta go
// to part [blocks][2][W]; column_reduce_kernel adds them (eight segments in block order, then the segments).
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const float* __restrict__ x, int64_t x_stride,
                                                                 const float* __restrict__ gamma, const float* __restrict__ dy,
                                                                 const float* __restrict__ dres, float* __restrict__ dx,
                                                                 float* __restrict__ part, int rows, int W, float eps,
                                                                 unsigned* __restrict__ amax_bits) {
    __shared__ float red[2][4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float am = 0.f;
    float4 dg[4], db[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) dg[t] = db[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int rr = 0; rr < LNB_ROWS / 4; ++rr) {
        const int row = blockIdx.x * LNB_ROWS + rr * 4 + wave;
        if (row >= rows) break;                                   // (wave-uniform)
        const float* src = x + (int64_t)row * x_stride;
        float4 v[4], g[4];
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = lane * 4 + t * 256;
            v[t] = w < W ? *reinterpret_cast<const float4*>(src + w) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
        }
        const float mean = cc_wave_sum_fast(s) / (float)W;
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = lane * 4 + t * 256;
            if (w < W) {
                v[t].x -= mean; v[t].y -= mean; v[t].z -= mean; v[t].w -= mean;
                q += (v[t].x * v[t].x + v[t].y * v[t].y) + (v[t].z * v[t].z + v[t].w * v[t].w);
            }
        }
        const float rstd = 1.0f / sqrtf(cc_wave_sum_fast(q) / (float)W + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = lane * 4 + t * 256;
            g[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (w < W) {
                v[t].x *= rstd; v[t].y *= rstd; v[t].z *= rstd; v[t].w *= rstd;          // xhat
                const float4 d = *reinterpret_cast<const float4*>(dy + (int64_t)row * W + w);
                const float4 gm = *reinterpret_cast<const float4*>(gamma + w);
                g[t] = make_float4(d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w);
                sg += (g[t].x + g[t].y) + (g[t].z + g[t].w);
                sgx += (g[t].x * v[t].x + g[t].y * v[t].y) + (g[t].z * v[t].z + g[t].w * v[t].w);
                dg[t].x += d.x * v[t].x; dg[t].y += d.y * v[t].y; dg[t].z += d.z * v[t].z; dg[t].w += d.w * v[t].w;
                db[t].x += d.x; db[t].y += d.y; db[t].z += d.z; db[t].w += d.w;
            }
        }
        const float c1 = cc_wave_sum_fast(sg) / (float)W, c2 = cc_wave_sum_fast(sgx) / (float)W;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = lane * 4 + t * 256;
            if (w < W) {
                float4 o = make_float4(rstd * (g[t].x - c1 - v[t].x * c2), rstd * (g[t].y - c1 - v[t].y * c2),
                                       rstd * (g[t].z - c1 - v[t].z * c2), rstd * (g[t].w - c1 - v[t].w * c2));
                if (dres) {
                    const float4 r = *reinterpret_cast<const float4*>(dres + (int64_t)row * W + w);
                    o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
                }
                *reinterpret_cast<float4*>(dx + (int64_t)row * W + w) = o;
                am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        *reinterpret_cast<float4*>(&red[0][wave][w]) = dg[t];
        *reinterpret_cast<float4*>(&red[1][wave][w]) = db[t];
    }
    __syncthreads();
    for (int w = threadIdx.x; w < W; w += 256) {
        part[((int64_t)blockIdx.x * 2 + 0) * W + w] = ((red[0][0][w] + red[0][1][w]) + red[0][2][w]) + red[0][3][w];
        part[((int64_t)blockIdx.x * 2 + 1) * W + w] = ((red[1][0][w] + red[1][1][w]) + red[1][2][w]) + red[1][3][w];
    }
    if (amax_bits) publish_absmax(am, amax_bits);
}

__global__ __launch_bounds__(256) void quick_gelu_backward_kernel(const _Float16* __restrict__ u_pre, const float* __restrict__ du,
                                                                  float* __restrict__ out, int64_t n,
                                                                  unsigned* __restrict__ amax_bits) {
    float am = 0.f;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        const h4 x = *reinterpret_cast<const h4*>(u_pre + i);
        const float4 d = *reinterpret_cast<const float4*>(du + i);
        float o[4];
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xv = (float)x[e];
            const float s = 1.0f / (1.0f + expf(-1.702f * xv));
            o[e] = dd[e] * (s + 1.702f * xv * s * (1.0f - s));
            am = fmaxf(am, fabsf(o[e]));
        }
        *reinterpret_cast<float4*>(out + i) = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (amax_bits) publish_absmax(am, amax_bits);
}

// u = QuickGELU(u_pre) on fp16 (the training forward keeps the pre-activation for the backward pass)
__global__ __launch_bounds__(256) void quick_gelu_f16_kernel(const _Float16* __restrict__ in, _Float16* __restrict__ out, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        const h4 x = *reinterpret_cast<const h4*>(in + i);
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xv = (float)x[e];
            o[e] = (_Float16)(xv / (1.0f + expf(-1.702f * xv)));
        }
        *reinterpret_cast<h4*>(out + i) = o;
    }
}

// column sums of [rows, cols] fp32: per-workgroup partials over row chunks in a fixed order, then one reduce
constexpr int CS_ROWS = 128;
__global__ __launch_bounds__(256) void column_partial_kernel(const float* __restrict__ in, int rows, int cols,
                                                             float* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cols) return;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += in[(int64_t)r * cols + c];
    part[(int64_t)blockIdx.y * cols + c] = s;
}
// out[c] = sum over chunks of part[chunk][c]: 32 columns x 8 chunk segments per workgroup, each segment added in chunk order and
// the eight segment sums in segment order (fixed bits; a three-workgroup serial loop over 300 partials took 27 us).  Columns
// >= split go to out2[c - split] (the LayerNorm backward's [blocks][2][W] partials: dgamma | dbeta).
__global__ __launch_bounds__(256) void column_reduce_kernel(const float* __restrict__ part, int chunks, int cols,
                                                            float* __restrict__ out, float* __restrict__ out2, int split) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int per = (chunks + 7) / 8, i0 = seg * per, i1 = min(chunks, i0 + per);
    float s = 0.f;
    if (c < cols)
        for (int i = i0; i < i1; ++i) s += part[(int64_t)i * cols + c];
    red[seg][cl] = s;
    __syncthreads();
    if (seg == 0 && c < cols) {
        float t = red[0][cl];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += red[k][cl];
        if (c < split) out[c] = t;
        else out2[c - split] = t;
    }
}

// Attention backward, one workgroup per (sequence, head), head_dim 64, L <= 64.  q, k, v rows of qkv [nseq*L, 3W] fp16
// (row = seq*L + token; heads are 64-wide slices), d_out [nseq*L, W] fp32 -> d_qkv [nseq*L, 3W] fp32.  Everything of a
// head lives in LDS as fp32; thread (i, c) loops are plain dot products in index order.
constexpr int AB_L = 64, AB_D = 64;
constexpr int AB_SMEM = (4 * AB_L * (AB_D + 1) + 2 * AB_L * (AB_L + 1)) * 4;
__global__ __launch_bounds__(256) void attention_backward_kernel(const _Float16* __restrict__ qkv, const float* __restrict__ d_out,
                                                                 float* __restrict__ d_qkv, int L, int heads, int W, int causal,
                                                                 unsigned* __restrict__ amax_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ab_smem[];          // AB_SMEM bytes (> the 64 KB static limit)
    float (*Q)[AB_D + 1] = reinterpret_cast<float (*)[AB_D + 1]>(ab_smem);
    float (*K)[AB_D + 1] = Q + AB_L;
    float (*V)[AB_D + 1] = K + AB_L;
    float (*dO)[AB_D + 1] = V + AB_L;
    float (*P)[AB_L + 1] = reinterpret_cast<float (*)[AB_L + 1]>(dO + AB_L);
    float (*dS)[AB_L + 1] = P + AB_L;
    const int seq = blockIdx.x / heads, head = blockIdx.x % heads, tid = threadIdx.x;
    const int64_t row0 = (int64_t)seq * L;
    for (int idx = tid; idx < L * AB_D; idx += 256) {
        const int t = idx / AB_D, d = idx % AB_D;
        const _Float16* r = qkv + (row0 + t) * 3 * W + head * AB_D + d;
        Q[t][d] = (float)r[0];
        K[t][d] = (float)r[W];
        V[t][d] = (float)r[2 * W];
        dO[t][d] = d_out[(row0 + t) * W + head * AB_D + d];
    }
    __syncthreads();
    // S = Q K^T / 8 (+ causal mask), P = softmax rows.  The products are register tiled - thread (bi, bj) owns the 4 x 4 block of
    // rows 4 bi.. and columns 4 bj..: 8 LDS reads per 16 multiply-adds instead of 2 per multiply-add (the kernel is bound by its
    // LDS reads); every element is still one dot product in d order.
    const int bi = tid >> 4, bj = tid & 15;                    // 16 x 16 blocks of 4 x 4 cover 64 x 64
    if (4 * bi < L && 4 * bj < L) {
        float acc[4][4] = {};
        for (int d = 0; d < AB_D; ++d) {
            float qa[4], kb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { qa[u] = Q[min(4 * bi + u, L - 1)][d]; kb[u] = K[min(4 * bj + u, L - 1)][d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(qa[u], kb[w], acc[u][w]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int i = 4 * bi + u, j = 4 * bj + w;
                if (i < L && j < L) P[i][j] = (causal && j > i) ? -INFINITY : acc[u][w] * 0.125f;
            }
    }
    __syncthreads();
    // the row passes: four threads per row (adjacent lanes), every fourth column each, combined by two lane exchanges - one
    // thread per row walked its 64 columns through LDS latency five times and was two thirds of the kernel
    const int rr = tid >> 2, rs = tid & 3;
    {
        float mx = -INFINITY;
        if (rr < L) for (int j = rs; j < L; j += 4) mx = fmaxf(mx, P[rr][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64)); mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sum = 0.f;
        if (rr < L) for (int j = rs; j < L; j += 4) { const float e = expf(P[rr][j] - mx); P[rr][j] = e; sum += e; }
        sum += __shfl_xor(sum, 1, 64); sum += __shfl_xor(sum, 2, 64);
        const float inv = 1.0f / sum;
        if (rr < L) for (int j = rs; j < L; j += 4) P[rr][j] *= inv;
    }
    __syncthreads();
    // dP = dO V^T ; dS = P (dP - sum_j dP P)
    if (4 * bi < L && 4 * bj < L) {
        float acc[4][4] = {};
        for (int d = 0; d < AB_D; ++d) {
            float oa[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { oa[u] = dO[min(4 * bi + u, L - 1)][d]; vb[u] = V[min(4 * bj + u, L - 1)][d]; }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int w = 0; w < 4; ++w) acc[u][w] = fmaf(oa[u], vb[w], acc[u][w]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int i = 4 * bi + u, j = 4 * bj + w;
                if (i < L && j < L) dS[i][j] = acc[u][w];
            }
    }
    __syncthreads();
    {
        float dot = 0.f;
        if (rr < L) for (int j = rs; j < L; j += 4) dot = fmaf(dS[rr][j], P[rr][j], dot);
        dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64);
        if (rr < L) for (int j = rs; j < L; j += 4) dS[rr][j] = P[rr][j] * (dS[rr][j] - dot);
    }
    __syncthreads();
    // dV = P^T dO ; dQ = dS K / 8 ; dK = dS^T Q / 8: thread (bt, bd) owns tokens 4 bt.. x features 4 bd..
    float am = 0.f;
    {
        const int bt = tid >> 4, bd = tid & 15;
        if (4 * bt < L) {
            float dv[4][4] = {}, dq[4][4] = {}, dk[4][4] = {};
            for (int j = 0; j < L; ++j) {
                float pj[4], sj[4], st[4], od[4], kd[4], qd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = min(4 * bt + u, L - 1);
                    pj[u] = P[j][t]; sj[u] = dS[j][t]; st[u] = dS[t][j];
                    od[u] = dO[j][4 * bd + u]; kd[u] = K[j][4 * bd + u]; qd[u] = Q[j][4 * bd + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        dv[u][w] = fmaf(pj[u], od[w], dv[u][w]);
                        dq[u][w] = fmaf(st[u], kd[w], dq[u][w]);
                        dk[u][w] = fmaf(sj[u], qd[w], dk[u][w]);
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = 4 * bt + u;
                if (t < L) {
                    float* o = d_qkv + (row0 + t) * 3 * W + head * AB_D + 4 * bd;
                    *reinterpret_cast<float4*>(o) = make_float4(dq[u][0] * 0.125f, dq[u][1] * 0.125f, dq[u][2] * 0.125f, dq[u][3] * 0.125f);
                    *reinterpret_cast<float4*>(o + W) = make_float4(dk[u][0] * 0.125f, dk[u][1] * 0.125f, dk[u][2] * 0.125f, dk[u][3] * 0.125f);
                    *reinterpret_cast<float4*>(o + 2 * W) = make_float4(dv[u][0], dv[u][1], dv[u][2], dv[u][3]);
#pragma unroll
                    for (int w = 0; w < 4; ++w)
                        am = fmaxf(fmaxf(am, fabsf(dv[u][w])), 0.125f * fmaxf(fabsf(dq[u][w]), fabsf(dk[u][w])));
                }
            }
        }
    }
    if (amax_bits) publish_absmax(am, amax_bits);
}

// Attention backward on the matrix cores (L <= 64, head_dim 64): the five 64 x 64 x 64 products of a head as fp16 MFMAs with fp32
// accumulators instead of fp32 FMAs (the fp32 kernel above spends 95 us per ViT-B/32 block on 3.7 GFLOP of VALU work; the
// tensors it moves are worth 30 us).  One workgroup per (sequence, head), wave w owns queries 16w..16w+15 for S / dP / dS / dQ
// and keys 16w..16w+15 for dV / dK.  fp16 operands: q, k, v as stored; dO scaled by a power of two chosen from the head's own
// largest |dO| (exact to apply and remove); P; dS scaled by a second power of two from the head's largest |dS| - the same
// operand precision the dgrad / wgrad GEMMs around it have.  MFMA convention (transformer.hip): D[x row 4 lg + e][y row l15] =
// sum_k X[x row][k] Y[y row][k], a fragment = 8 consecutive k of row (lane & 15) at k = 32 ks + 8 (lane >> 4).
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ABM_S = 72;                                        // LDS row stride (halves): 144 B rows, 16-byte aligned
constexpr int ABM_SMEM = (6 * 64 + 4 * 16) * ABM_S * 2 + 64;
__device__ __forceinline__ float pow2_scale_for(float amax) {    // 2^k with 2^k amax in [8192, 16384): cc_cast_scaled_f16's rule
    return (amax > 0.f && isfinite(amax)) ? exp2f(fminf(floorf(log2f(16384.0f / amax)), 100.f)) : 1.0f;
}
__global__ __launch_bounds__(256) void attention_backward_mfma_kernel(const _Float16* __restrict__ qkv, const float* __restrict__ d_out,
                                                                      float* __restrict__ d_qkv, int L, int heads, int W, int causal,
                                                                      unsigned* __restrict__ amax_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char abm_smem[];
    _Float16* Kt = reinterpret_cast<_Float16*>(abm_smem);        // [d][token]
    _Float16* Qt = Kt + 64 * ABM_S;                              // [d][token]
    _Float16* dOt = Qt + 64 * ABM_S;                             // [d][token]   (scaled)
    _Float16* dO16 = dOt + 64 * ABM_S;                           // [token][d]   (scaled)
    _Float16* Pt = dO16 + 64 * ABM_S;                            // [key][query]
    _Float16* dSt = Pt + 64 * ABM_S;                             // [key][query] (scaled)
    _Float16* dSw = dSt + 64 * ABM_S;                            // per wave [16 queries][keys] (scaled)
    float* wred = reinterpret_cast<float*>(dSw + 4 * 16 * ABM_S);      // [8]: two cross-wave maxima
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int seq = blockIdx.x / heads, head = blockIdx.x - seq * heads;
    const int64_t row0 = (int64_t)seq * L;
    const int64_t ld = 3 * (int64_t)W;
    const _Float16* base = qkv + row0 * ld + head * 64;
    const float* dob = d_out + row0 * W + head * 64;

    // ---- stage: K^T, Q^T (fp16 as stored) and dO (fp32 -> scaled fp16, both layouts); token rows >= L are zeros
    float4 dof[2][2];
    h8 kc[2], qc[2];
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
        if (r < L) {
            kc[c] = *reinterpret_cast<const h8*>(base + (int64_t)r * ld + W + ch * 8);
            qc[c] = *reinterpret_cast<const h8*>(base + (int64_t)r * ld + ch * 8);
            dof[c][0] = *reinterpret_cast<const float4*>(dob + (int64_t)r * W + ch * 8);
            dof[c][1] = *reinterpret_cast<const float4*>(dob + (int64_t)r * W + ch * 8 + 4);
        } else {
            kc[c] = qc[c] = h8{0, 0, 0, 0, 0, 0, 0, 0};
            dof[c][0] = dof[c][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            am = fmaxf(fmaxf(am, fmaxf(fabsf(dof[c][h].x), fabsf(dof[c][h].y))), fmaxf(fabsf(dof[c][h].z), fabsf(dof[c][h].w)));
    }
    am = cc_wave_max(am);
    if (lane == 0) wred[wave] = am;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            Kt[(ch * 8 + e) * ABM_S + r] = kc[c][e];
            Qt[(ch * 8 + e) * ABM_S + r] = qc[c][e];
        }
    }
    __syncthreads();
    const float s_o = pow2_scale_for(fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
        const float v[8] = {dof[c][0].x, dof[c][0].y, dof[c][0].z, dof[c][0].w, dof[c][1].x, dof[c][1].y, dof[c][1].z, dof[c][1].w};
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = (_Float16)(v[e] * s_o);
            dOt[(ch * 8 + e) * ABM_S + r] = o[e];
        }
        *reinterpret_cast<h8*>(dO16 + r * ABM_S + ch * 8) = o;
    }
    // ---- the wave's query strip: S = Q K^T / 8 -> P, dP = dO V^T (operands of the key side straight from global memory)
    const int qi = wave * 16 + l15;                               // the query this lane's accumulators belong to
    const int qrow = min(qi, L - 1);
    h8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const h8*>(base + (int64_t)qrow * ld + (ks * 4 + lg) * 8);
    f32x4 p[4];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int kr = min(kt * 16 + l15, L - 1);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const h8 kf = *reinterpret_cast<const h8*>(base + (int64_t)kr * ld + W + (ks * 4 + lg) * 8);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], a, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kt * 16 + lg * 4 + e;
            const bool ok = key < L && (!causal || key <= qi);
            a[e] = ok ? a[e] * 0.125f : -3.0e38f;
            mx = fmaxf(mx, a[e]);
        }
        p[kt] = a;
    }
    mx = cc_rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pe = (p[kt][e] > -1.0e38f) ? __expf(p[kt][e] - mx) : 0.f;
            p[kt][e] = pe;
            sum += pe;
        }
    sum = cc_rows_sum(sum);
    const float inv = (qi < L) ? 1.0f / sum : 0.f;                // (padding queries contribute nothing)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) p[kt][e] *= inv;
    __syncthreads();                                              // dO16 / dOt / Kt / Qt complete
    f32x4 ds[4];
    float dot = 0.f;
    {
        h8 of[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) of[ks] = *reinterpret_cast<const h8*>(dO16 + qi * ABM_S + (ks * 4 + lg) * 8);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const int kr = min(kt * 16 + l15, L - 1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 vf = *reinterpret_cast<const h8*>(base + (int64_t)kr * ld + 2 * W + (ks * 4 + lg) * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, of[ks], a, 0, 0, 0);
            }
            ds[kt] = a;                                           // dP (times s_o)
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = fmaf(a[e], p[kt][e], dot);
        }
    }
    dot = cc_rows_sum(dot);
    float ams = 0.f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ds[kt][e] = p[kt][e] * (ds[kt][e] - dot);             // dS (times s_o); 0 where P is 0
            ams = fmaxf(ams, fabsf(ds[kt][e]));
        }
    ams = cc_wave_max(ams);
    if (lane == 0) wred[4 + wave] = ams;
    __syncthreads();
    const float s_s = pow2_scale_for(fmaxf(fmaxf(wred[4], wred[5]), fmaxf(wred[6], wred[7])));
    _Float16* dSm = dSw + wave * 16 * ABM_S;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        h4 d4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kt * 16 + lg * 4 + e;
            d4[e] = (_Float16)(ds[kt][e] * s_s);
            Pt[key * ABM_S + qi] = (_Float16)p[kt][e];
            dSt[key * ABM_S + qi] = d4[e];
        }
        *reinterpret_cast<h4*>(dSm + l15 * ABM_S + kt * 16 + lg * 4) = d4;
    }
    __syncthreads();
    // ---- dQ (this wave's queries) = dS K / 8;  dV, dK (this wave's keys) = P^T dO, dS^T Q / 8
    const float un_v = 1.0f / s_o, un_q = 0.125f * un_v / s_s;
    const bool live = qi < L;                                     // qi doubles as the key index of dV / dK rows
    float* orow = d_qkv + (row0 + qi) * ld + head * 64;
    float amo = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f}, av = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
        const int d = dt * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const int ko = kb * 32 + lg * 8;
            const h8 ktf = *reinterpret_cast<const h8*>(Kt + d * ABM_S + ko);
            const h8 dsf = *reinterpret_cast<const h8*>(dSm + l15 * ABM_S + ko);
            aq = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf, dsf, aq, 0, 0, 0);
            const h8 otf = *reinterpret_cast<const h8*>(dOt + d * ABM_S + ko);
            const h8 ptf = *reinterpret_cast<const h8*>(Pt + qi * ABM_S + ko);
            av = __builtin_amdgcn_mfma_f32_16x16x32_f16(otf, ptf, av, 0, 0, 0);
            const h8 qtf = *reinterpret_cast<const h8*>(Qt + d * ABM_S + ko);
            const h8 stf = *reinterpret_cast<const h8*>(dSt + qi * ABM_S + ko);
            ak = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtf, stf, ak, 0, 0, 0);
        }
        if (live) {
            const float4 oq = make_float4(aq[0] * un_q, aq[1] * un_q, aq[2] * un_q, aq[3] * un_q);
            const float4 ok_ = make_float4(ak[0] * un_q, ak[1] * un_q, ak[2] * un_q, ak[3] * un_q);
            const float4 ov = make_float4(av[0] * un_v, av[1] * un_v, av[2] * un_v, av[3] * un_v);
            float* o = orow + dt * 16 + lg * 4;
            *reinterpret_cast<float4*>(o) = oq;
            *reinterpret_cast<float4*>(o + W) = ok_;
            *reinterpret_cast<float4*>(o + 2 * W) = ov;
            amo = fmaxf(amo, fmaxf(fmaxf(fmaxf(fabsf(oq.x), fabsf(oq.y)), fmaxf(fabsf(oq.z), fabsf(oq.w))),
                                   fmaxf(fmaxf(fabsf(ok_.x), fabsf(ok_.y)), fmaxf(fabsf(ok_.z), fabsf(ok_.w)))));
            amo = fmaxf(amo, fmaxf(fmaxf(fabsf(ov.x), fabsf(ov.y)), fmaxf(fabsf(ov.z), fabsf(ov.w))));
        }
    }
    if (amax_bits) publish_absmax(amo, amax_bits);
}

// ---- 64 < L <= 256 (ViT-B/16: 197 tokens per frame, 161 in its clustered blocks): the same arithmetic as two launches.
//   q side: one workgroup per (sequence, head, 64 queries); wave w holds its 16 queries' rows of S, P, dP and dS against ALL keys in
//           registers (16 accumulators each at L = 256), writes dQ and, per query, log-sum-exp and D = sum_j P dP to `stats`
//   k side: one workgroup per (sequence, head, 64 keys); loops over the query tiles, rebuilds P^T = exp(S^T / 8 - lse) and
//           dS^T = P^T (dP^T - D) for its keys from those two numbers and accumulates dV = P^T dO, dK = dS^T Q / 8
// Scales: dO by a power of two from the query tile's largest |dO| (q side: workgroup; k side: per query tile, divided out of each
// tile's product before it is added), dS by a power of two from the WAVE's largest |dS| (both contractions that consume a wave's
// dS are that wave's own).
template <int NK64>
__global__ __launch_bounds__(256) void attention_backward_q_kernel(const _Float16* __restrict__ qkv, const float* __restrict__ d_out,
                                                                   float* __restrict__ d_qkv, float* __restrict__ stats, int L, int heads,
                                                                   int W, int causal, unsigned* __restrict__ amax_bits) {
    constexpr int LP = NK64 * 64, NKT = NK64 * 4, KS = LP + 8;
    extern __shared__ __attribute__((aligned(16))) unsigned char abq_smem[];
    _Float16* Kt = reinterpret_cast<_Float16*>(abq_smem);        // [d][key]
    _Float16* dO16 = Kt + 64 * KS;                               // [query of the tile][d] (scaled)
    _Float16* dSw = dO16 + 64 * ABM_S;                           // per wave [16 queries][keys] (scaled)
    float* wred = reinterpret_cast<float*>(dSw + 4 * 16 * KS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int QT = (L + 63) >> 6;
    const int qt = blockIdx.x % QT, sh = blockIdx.x / QT, seq = sh / heads, head = sh - seq * heads;
    const int64_t row0 = (int64_t)seq * L, ld = 3 * (int64_t)W;
    const _Float16* base = qkv + row0 * ld + head * 64;
    const float* dob = d_out + row0 * W + head * 64;
    // ---- stage K^T (all keys) and this tile's dO
#pragma unroll
    for (int c = 0; c < NK64 * 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
        const h8 kc = r < L ? *reinterpret_cast<const h8*>(base + (int64_t)r * ld + W + ch * 8) : h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) Kt[(ch * 8 + e) * KS + r] = kc[e];
    }
    float4 dof[2][2];
    float am = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7, gi = qt * 64 + r;
        if (gi < L) {
            dof[c][0] = *reinterpret_cast<const float4*>(dob + (int64_t)gi * W + ch * 8);
            dof[c][1] = *reinterpret_cast<const float4*>(dob + (int64_t)gi * W + ch * 8 + 4);
        } else {
            dof[c][0] = dof[c][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            am = fmaxf(fmaxf(am, fmaxf(fabsf(dof[c][h].x), fabsf(dof[c][h].y))), fmaxf(fabsf(dof[c][h].z), fabsf(dof[c][h].w)));
    }
    am = cc_wave_max(am);
    if (lane == 0) wred[wave] = am;
    __syncthreads();
    const float s_o = pow2_scale_for(fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
        const float v[8] = {dof[c][0].x, dof[c][0].y, dof[c][0].z, dof[c][0].w, dof[c][1].x, dof[c][1].y, dof[c][1].z, dof[c][1].w};
        h8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(v[e] * s_o);
        *reinterpret_cast<h8*>(dO16 + r * ABM_S + ch * 8) = o;
    }
    // ---- S row strip -> P
    const int ql = wave * 16 + l15, qi = qt * 64 + ql;
    const int qrow = min(qi, L - 1);
    h8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const h8*>(base + (int64_t)qrow * ld + (ks * 4 + lg) * 8);
    f32x4 p[NKT];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const int kr = min(kt * 16 + l15, L - 1);
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const h8 kf = *reinterpret_cast<const h8*>(base + (int64_t)kr * ld + W + (ks * 4 + lg) * 8);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], a, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = kt * 16 + lg * 4 + e;
            const bool ok = key < L && (!causal || key <= qi);
            a[e] = ok ? a[e] * 0.125f : -3.0e38f;
            mx = fmaxf(mx, a[e]);
        }
        p[kt] = a;
    }
    mx = cc_rows_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float pe = (p[kt][e] > -1.0e38f) ? __expf(p[kt][e] - mx) : 0.f;
            p[kt][e] = pe;
            sum += pe;
        }
    sum = cc_rows_sum(sum);
    const bool liveq = qi < L;
    const float inv = liveq ? 1.0f / sum : 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) p[kt][e] *= inv;
    __syncthreads();                                              // dO16 and Kt complete
    // ---- dP strip, D, dS
    f32x4 ds[NKT];
    float dot = 0.f;
    {
        h8 of[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) of[ks] = *reinterpret_cast<const h8*>(dO16 + ql * ABM_S + (ks * 4 + lg) * 8);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            const int kr = min(kt * 16 + l15, L - 1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 vf = *reinterpret_cast<const h8*>(base + (int64_t)kr * ld + 2 * W + (ks * 4 + lg) * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, of[ks], a, 0, 0, 0);
            }
            ds[kt] = a;
#pragma unroll
            for (int e = 0; e < 4; ++e) dot = fmaf(a[e], p[kt][e], dot);
        }
    }
    dot = cc_rows_sum(dot);
    if (liveq && lg == 0) {
        float* st = stats + ((row0 + qi) * heads + head) * 2;
        st[0] = mx + __logf(sum);                                 // log-sum-exp of the scaled, masked scores
        st[1] = dot / s_o;                                        // D = sum_j P dP (unscaled)
    }
    float ams = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            ds[kt][e] = p[kt][e] * (ds[kt][e] - dot);
            ams = fmaxf(ams, fabsf(ds[kt][e]));
        }
    const float s_s = pow2_scale_for(cc_wave_max(ams));           // per wave: dQ of these 16 queries contracts this strip only
    _Float16* dSm = dSw + wave * 16 * KS;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
        const h4 d4 = {(_Float16)(ds[kt][0] * s_s), (_Float16)(ds[kt][1] * s_s), (_Float16)(ds[kt][2] * s_s), (_Float16)(ds[kt][3] * s_s)};
        *reinterpret_cast<h4*>(dSm + l15 * KS + kt * 16 + lg * 4) = d4;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    // ---- dQ = dS K / 8
    const float un_q = 0.125f / s_o / s_s;
    float amo = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        f32x4 aq = {0.f, 0.f, 0.f, 0.f};
        const int d = dt * 16 + l15;
#pragma unroll
        for (int kb = 0; kb < LP / 32; ++kb) {
            const int ko = kb * 32 + lg * 8;
            aq = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(Kt + d * KS + ko),
                                                        *reinterpret_cast<const h8*>(dSm + l15 * KS + ko), aq, 0, 0, 0);
        }
        if (liveq) {
            const float4 oq = make_float4(aq[0] * un_q, aq[1] * un_q, aq[2] * un_q, aq[3] * un_q);
            *reinterpret_cast<float4*>(d_qkv + (row0 + qi) * ld + head * 64 + dt * 16 + lg * 4) = oq;
            amo = fmaxf(amo, fmaxf(fmaxf(fabsf(oq.x), fabsf(oq.y)), fmaxf(fabsf(oq.z), fabsf(oq.w))));
        }
    }
    if (amax_bits) publish_absmax(amo, amax_bits);
}

constexpr int ABK_SMEM = 5 * 64 * ABM_S * 2 + (128 + 8) * 4;
__global__ __launch_bounds__(256) void attention_backward_kv_kernel(const _Float16* __restrict__ qkv, const float* __restrict__ d_out,
                                                                    float* __restrict__ d_qkv, const float* __restrict__ stats, int L,
                                                                    int heads, int W, int causal, unsigned* __restrict__ amax_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char abk_smem[];
    _Float16* Qt = reinterpret_cast<_Float16*>(abk_smem);        // [d][query of the tile]
    _Float16* dOt = Qt + 64 * ABM_S;                             // [d][query]   (scaled)
    _Float16* dO16 = dOt + 64 * ABM_S;                           // [query][d]   (scaled)
    _Float16* Pt = dO16 + 64 * ABM_S;                            // [key of the tile][query]   (rows 16w.. are wave w's own)
    _Float16* dSt = Pt + 64 * ABM_S;                             // [key][query] (scaled per wave)
    float* lse = reinterpret_cast<float*>(dSt + 64 * ABM_S);     // [64] + D [64]
    float* dd = lse + 64;
    float* wred = dd + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const int QT = (L + 63) >> 6;
    const int kt64 = blockIdx.x % QT, sh = blockIdx.x / QT, seq = sh / heads, head = sh - seq * heads;
    const int64_t row0 = (int64_t)seq * L, ld = 3 * (int64_t)W;
    const _Float16* base = qkv + row0 * ld + head * 64;
    const float* dob = d_out + row0 * W + head * 64;
    const int kl = wave * 16 + l15, kj = kt64 * 64 + kl;          // this lane's key (the y row of every product below)
    const int krow = min(kj, L - 1);
    h8 kf[2], vf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        kf[ks] = *reinterpret_cast<const h8*>(base + (int64_t)krow * ld + W + (ks * 4 + lg) * 8);
        vf[ks] = *reinterpret_cast<const h8*>(base + (int64_t)krow * ld + 2 * W + (ks * 4 + lg) * 8);
    }
    f32x4 dv[4], dk[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dv[dt] = dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int q_end = causal ? QT : QT;                           // (causal: tiles behind the key tile are all masked but cost little)
    for (int qt = 0; qt < q_end; ++qt) {
        __syncthreads();                                          // the previous tile's operands are no longer read
        float4 dof[2][2];
        h8 qc[2];
        float am = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7, gi = qt * 64 + r;
            if (gi < L) {
                qc[c] = *reinterpret_cast<const h8*>(base + (int64_t)gi * ld + ch * 8);
                dof[c][0] = *reinterpret_cast<const float4*>(dob + (int64_t)gi * W + ch * 8);
                dof[c][1] = *reinterpret_cast<const float4*>(dob + (int64_t)gi * W + ch * 8 + 4);
            } else {
                qc[c] = h8{0, 0, 0, 0, 0, 0, 0, 0};
                dof[c][0] = dof[c][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                am = fmaxf(fmaxf(am, fmaxf(fabsf(dof[c][h].x), fabsf(dof[c][h].y))), fmaxf(fabsf(dof[c][h].z), fabsf(dof[c][h].w)));
#pragma unroll
            for (int e = 0; e < 8; ++e) Qt[(ch * 8 + e) * ABM_S + r] = qc[c][e];
        }
        am = cc_wave_max(am);
        if (lane == 0) wred[wave] = am;
        if (tid < 64) {
            const int gi = qt * 64 + tid;
            const float* st = stats + ((row0 + min(gi, L - 1)) * heads + head) * 2;
            lse[tid] = st[0];
            dd[tid] = st[1];
        }
        __syncthreads();
        const float s_o = pow2_scale_for(fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3])));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int idx = c * 256 + tid, r = idx >> 3, ch = idx & 7;
            const float v[8] = {dof[c][0].x, dof[c][0].y, dof[c][0].z, dof[c][0].w, dof[c][1].x, dof[c][1].y, dof[c][1].z, dof[c][1].w};
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = (_Float16)(v[e] * s_o);
                dOt[(ch * 8 + e) * ABM_S + r] = o[e];
            }
            *reinterpret_cast<h8*>(dO16 + r * ABM_S + ch * 8) = o;
        }
        __syncthreads();
        // ---- P^T and dS^T of (this wave's 16 keys) x (the tile's 64 queries): lane = key l15, queries it*16 + 4 lg + e
        f32x4 pt[4], dst[4];
        float ams = 0.f;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int qr = min(qt * 64 + it * 16 + l15, L - 1);
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const h8 qf = *reinterpret_cast<const h8*>(base + (int64_t)qr * ld + (ks * 4 + lg) * 8);
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(qf, kf[ks], a, 0, 0, 0);
                const h8 of = *reinterpret_cast<const h8*>(dO16 + (it * 16 + l15) * ABM_S + (ks * 4 + lg) * 8);
                b = __builtin_amdgcn_mfma_f32_16x16x32_f16(of, vf[ks], b, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int il = it * 16 + lg * 4 + e, gi = qt * 64 + il;
                const bool ok = gi < L && kj < L && (!causal || kj <= gi);
                const float pe = ok ? __expf(a[e] * 0.125f - lse[il]) : 0.f;
                pt[it][e] = pe;
                dst[it][e] = pe * (b[e] - dd[il] * s_o);
                ams = fmaxf(ams, fabsf(dst[it][e]));
            }
        }
        const float s_s = pow2_scale_for(cc_wave_max(ams));
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const h4 p4 = {(_Float16)pt[it][0], (_Float16)pt[it][1], (_Float16)pt[it][2], (_Float16)pt[it][3]};
            const h4 d4 = {(_Float16)(dst[it][0] * s_s), (_Float16)(dst[it][1] * s_s), (_Float16)(dst[it][2] * s_s), (_Float16)(dst[it][3] * s_s)};
            *reinterpret_cast<h4*>(Pt + kl * ABM_S + it * 16 + lg * 4) = p4;
            *reinterpret_cast<h4*>(dSt + kl * ABM_S + it * 16 + lg * 4) = d4;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- dV += P^T dO / s_o ; dK += dS^T Q / (8 s_o s_s)
        const float un_v = 1.0f / s_o, un_k = 0.125f * un_v / s_s;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            f32x4 av = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
            const int d = dt * 16 + l15;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                const int ko = kb * 32 + lg * 8;
                av = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(dOt + d * ABM_S + ko),
                                                            *reinterpret_cast<const h8*>(Pt + kl * ABM_S + ko), av, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const h8*>(Qt + d * ABM_S + ko),
                                                            *reinterpret_cast<const h8*>(dSt + kl * ABM_S + ko), ak, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dv[dt][e] = fmaf(av[e], un_v, dv[dt][e]);
                dk[dt][e] = fmaf(ak[e], un_k, dk[dt][e]);
            }
        }
    }
    float amo = 0.f;
    if (kj < L) {
        float* o = d_qkv + (row0 + kj) * ld + head * 64 + lg * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            *reinterpret_cast<float4*>(o + W + dt * 16) = make_float4(dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]);
            *reinterpret_cast<float4*>(o + 2 * W + dt * 16) = make_float4(dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) amo = fmaxf(amo, fmaxf(fabsf(dk[dt][e]), fabsf(dv[dt][e])));
        }
    }
    if (amax_bits) publish_absmax(amo, amax_bits);
}

// ---- fp32 gradients through the fp16 matrix cores: |x| max -> scale = 2^k with scale * max in [8192, 16384) -> fp16 copy
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ in, int64_t n, unsigned* __restrict__ out_bits) {
    float m = 0.f;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(in) & 15) == 0) ? (n >> 2) : 0;     // (vector loads where the view is aligned)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(in[i]));
    m = cc_wave_max(m);
    // one atomic per workgroup (28.8 k wave atomics on one address took three times as long as reading the tensor)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned mine = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));   // (non-negative floats order as their bits)
        if (mine > __atomic_load_n(out_bits, __ATOMIC_RELAXED)) atomicMax(out_bits, mine);
    }
}
__global__ __launch_bounds__(256) void cast_scaled_kernel(const float* __restrict__ in, _Float16* __restrict__ out, int64_t n,
                                                          const float* __restrict__ amax, float* __restrict__ scale_out) {
    const float a = *amax;
    // 2^k: exact to apply and to remove
    const float scale = (a > 0.f && isfinite(a)) ? exp2f(floorf(log2f(16384.0f / a))) : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) *scale_out = scale;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * 1024) {
        if (i + 3 < n) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            const h4 o = {(_Float16)(v.x * scale), (_Float16)(v.y * scale), (_Float16)(v.z * scale), (_Float16)(v.w * scale)};
            *reinterpret_cast<h4*>(out + i) = o;
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = (_Float16)(in[j] * scale);
        }
    }
}
// The same cast for a matrix in [rows, cols], writing the fp16 copy row-major AND transposed [cols, rows_pad] (zero columns
// behind `rows`): the two operand layouts a Linear's backward multiplies (dX = dY W: dY row-major; dW = dY^T X: both operands
// with the row count as the contraction, cc_linear_f16 wants it contiguous and a multiple of 64).  One read of the fp32
// matrix, 64 x 64 tiles through LDS.  in_f16 != null: the source is already fp16 (activations saved by the forward), no scale.
// amax == null: scale 1 (weights).
typedef _Float16 bh8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ in, const _Float16* __restrict__ in_f16,
                                                             _Float16* __restrict__ out, _Float16* __restrict__ out_t, int rows,
                                                             int cols, int rows_pad, const float* __restrict__ amax,
                                                             float* __restrict__ scale_out, float* __restrict__ col_partial) {
    __shared__ _Float16 tile[64][72];
    __shared__ float csum[16][64];                                // col_partial: column sums of the tile's (unscaled) fp32 rows
    float scale = 1.f;
    if (amax) {
        const float a = *amax;
        scale = (a > 0.f && isfinite(a)) ? exp2f(floorf(log2f(16384.0f / a))) : 1.0f;
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && scale_out) *scale_out = scale;
    }
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 4; ++q) {                                 // 64 rows x 16 column quads
        const int idx = q * 256 + tid, r = idx >> 4, c = (idx & 15) * 4;
        const int gr = r0 + r, gc = c0 + c;
        h4 o = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
        if (gr < rows && gc < cols) {                             // (cols % 4 == 0)
            if (in_f16) {
                o = *reinterpret_cast<const h4*>(in_f16 + (int64_t)gr * cols + gc);
            } else {
                const float4 v = *reinterpret_cast<const float4*>(in + (int64_t)gr * cols + gc);
                o = h4{(_Float16)(v.x * scale), (_Float16)(v.y * scale), (_Float16)(v.z * scale), (_Float16)(v.w * scale)};
                cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;                  // rows q*16 + tid/16, ascending in q
            }
            if (out) *reinterpret_cast<h4*>(out + (int64_t)gr * cols + gc) = o;
        }
        *reinterpret_cast<h4*>(&tile[r][c]) = o;
    }
    if (col_partial) *reinterpret_cast<float4*>(&csum[tid >> 4][(tid & 15) * 4]) = cs;
    __syncthreads();
    if (col_partial && tid < 64 && c0 + tid < cols) {             // fixed order: the 16 row groups of the tile, ascending
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += csum[u][tid];
        col_partial[(int64_t)blockIdx.y * cols + c0 + tid] = t;
    }
    if (!out_t) return;                                           // (round 5: the weight gradient reads dY and X as they lie)
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                 // 64 columns x 8 row octets
        const int idx = q * 256 + tid, c = idx >> 3, r = (idx & 7) * 8;
        const int gc = c0 + c, gr = r0 + r;
        if (gc < cols && gr < rows_pad) {                         // (rows_pad % 8 == 0)
            bh8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = tile[r + e][c];
            *reinterpret_cast<bh8*>(out_t + (int64_t)gc * rows_pad + gr) = o;
        }
    }
}
__global__ __launch_bounds__(256) void unscale_kernel(float* __restrict__ x, int64_t n, const float* __restrict__ sa,
                                                      const float* __restrict__ sb) {
    const float inv = 1.0f / ((*sa) * (sb ? *sb : 1.0f));
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) x[i] *= inv;
}

inline unsigned grid_for(int64_t n, int per_block) {
    const int64_t b = (n + per_block - 1) / per_block;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace

// ---------------------------------------------------------------------------------------------------- BertAdam
// utils/optimization.py:100-170, one parameter tensor per call: per-tensor gradient clipping (clip_grad_norm_(p, max_grad_norm):
// g *= max / (||g|| + 1e-6) when that factor is < 1 - the gradient tensor is rescaled in place as the reference does), moments,
// update = m / (sqrt(v) + e) [+ weight_decay * p], p -= lr_scheduled * update.  Two launches: the sum of squares (a fixed
// grid, double atomics would make it order dependent: per-block partials reduced in block order by the second kernel), then
// the elementwise step.  No bias correction - this is the BERT variant.
constexpr int BA_BLOCKS = 512;
__global__ __launch_bounds__(256) void bertadam_norm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ partial) {
    double s = 0.0;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? (n >> 2) : 0;       // (gradients may be views of a flat bucket)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double v = (double)g[i];
        s += v * v;
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void bertadam_step_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                             float* __restrict__ v, int64_t n, const double* __restrict__ partial,
                                                             int nblocks, float lr, float b1, float b2, float eps, float wd,
                                                             float max_norm, const float* __restrict__ lr_dev) {
    if (lr_dev) lr = *lr_dev;                                      // (a captured step: the schedule's value arrives through memory)
    float coef = 1.f;
    if (max_norm > 0.f) {
        // the block partials, summed by the first wave in a fixed order (lane l takes l, l + 64, ...; then the wave tree)
        __shared__ double tot_s;
        if (threadIdx.x < 64) {
            double t = 0.0;
            for (int b = threadIdx.x; b < nblocks; b += 64) t += partial[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (threadIdx.x == 0) tot_s = t;
        }
        __syncthreads();
        const float c = max_norm / ((float)sqrt(tot_s) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float upd = mi / (sqrtf(vi) + eps);
        const float pi = p[i];
        if (wd > 0.f) upd += wd * pi;
        g[i] = gi; m[i] = mi; v[i] = vi;
        p[i] = pi - lr * upd;
    }
}
// small tensors (biases, LayerNorm parameters: two thirds of a CLIP model's tensors): norm and step by ONE workgroup
__device__ __forceinline__ void bertadam_small_body(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int n, float lr, float b1, float b2, float eps, float wd,
                                                    float max_norm) {
    float coef = 1.f;
    if (max_norm > 0.f) {
        __shared__ double red[4];
        double s = 0.0;
        for (int i = threadIdx.x; i < n; i += 256) { const double x = (double)g[i]; s += x * x; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        const float c = max_norm / ((float)sqrt((red[0] + red[1]) + (red[2] + red[3])) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    for (int i = threadIdx.x; i < n; i += 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float upd = mi / (sqrtf(vi) + eps);
        const float pi = p[i];
        if (wd > 0.f) upd += wd * pi;
        g[i] = gi; m[i] = mi; v[i] = vi;
        p[i] = pi - lr * upd;
    }
}
__global__ __launch_bounds__(256) void bertadam_small_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                              float* __restrict__ v, int n, float lr, float b1, float b2, float eps,
                                                              float wd, float max_norm, const float* __restrict__ lr_dev) {
    if (lr_dev) lr = *lr_dev;
    bertadam_small_body(p, g, m, v, n, lr, b1, b2, eps, wd, max_norm);
}
// ... and all of a model's small tensors in one launch: workgroup i takes items[i] (cc_bertadam_item, include/centerclip_hip.h)
struct BertAdamItem {
    float* p; float* g; float* m; float* v;
    const float* lr_dev;
    int32_t n;
    float lr, wd;
    int32_t pad_;
};
static_assert(sizeof(BertAdamItem) == 56, "cc_bertadam_item layout");
// ... and all LARGE tensors in two launches (cc_bertadam_multi_large_f32): the workgroups of every tensor's norm pass, then the
// workgroups of every tensor's step, each finding its tensor by bisection over the records' first-block numbers.  A tensor's
// workgroups see the block count and block id cc_bertadam_step_f32 would give them, so its arithmetic and bits are the same.
struct BertAdamBigItem {
    float* p; float* g; float* m; float* v;
    const float* lr_dev;
    int64_t n;
    float lr, wd;
    int32_t norm_blk0, norm_blocks, step_blk0, step_blocks;
};
static_assert(sizeof(BertAdamBigItem) == 72, "cc_bertadam_big_item layout");
template <bool STEP>
__device__ __forceinline__ int bertadam_find(const BertAdamBigItem* __restrict__ items, int count, int blk) {
    int lo = 0, hi = count - 1;                                   // last item whose first block <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((STEP ? items[mid].step_blk0 : items[mid].norm_blk0) <= blk) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ __launch_bounds__(256) void bertadam_multi_norm_kernel(const BertAdamBigItem* __restrict__ items, int count,
                                                                   double* __restrict__ partial) {
    const int ii = bertadam_find<false>(items, count, blockIdx.x);
    const float* __restrict__ g = items[ii].g;
    const int64_t n = items[ii].n;
    const int64_t bid = (int)blockIdx.x - items[ii].norm_blk0, nblk = items[ii].norm_blocks;
    double s = 0.0;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(g) & 15) == 0) ? (n >> 2) : 0;
    for (int64_t i = bid * 256 + threadIdx.x; i < n4; i += nblk * 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    for (int64_t i = (n4 << 2) + bid * 256 + threadIdx.x; i < n; i += nblk * 256) {
        const double v = (double)g[i];
        s += v * v;
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void bertadam_multi_step_kernel(const BertAdamBigItem* __restrict__ items, int count,
                                                                   const double* __restrict__ partial, float b1, float b2, float eps,
                                                                   float max_norm) {
    const int ii = bertadam_find<true>(items, count, blockIdx.x);
    const BertAdamBigItem it = items[ii];
    const float lr = it.lr_dev ? *it.lr_dev : it.lr;
    float coef = 1.f;
    if (max_norm > 0.f) {
        __shared__ double tot_s;
        if (threadIdx.x < 64) {
            double t = 0.0;
            for (int b = threadIdx.x; b < it.norm_blocks; b += 64) t += partial[it.norm_blk0 + b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
            if (threadIdx.x == 0) tot_s = t;
        }
        __syncthreads();
        const float c = max_norm / ((float)sqrt(tot_s) + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    float* __restrict__ p = it.p; float* __restrict__ g = it.g; float* __restrict__ m = it.m; float* __restrict__ v = it.v;
    const int64_t bid = (int)blockIdx.x - it.step_blk0;
    for (int64_t i = bid * 256 + threadIdx.x; i < it.n; i += (int64_t)it.step_blocks * 256) {
        const float gi = g[i] * coef;
        const float mi = m[i] * b1 + (1.f - b1) * gi;
        const float vi = v[i] * b2 + (1.f - b2) * gi * gi;
        float upd = mi / (sqrtf(vi) + eps);
        const float pi = p[i];
        if (it.wd > 0.f) upd += it.wd * pi;
        g[i] = gi; m[i] = mi; v[i] = vi;
        p[i] = pi - lr * upd;
    }
}
__global__ __launch_bounds__(256) void bertadam_multi_small_kernel(const BertAdamItem* __restrict__ items, float b1, float b2, float eps,
                                                                    float max_norm) {
    const BertAdamItem it = items[blockIdx.x];
    bertadam_small_body(it.p, it.g, it.m, it.v, it.n, it.lr_dev ? *it.lr_dev : it.lr, b1, b2, eps, it.wd, max_norm);
}

template <int NK64>
static int launch_attention_backward_long(const _Float16* qkv, const float* d_out, float* d_qkv, float* stats, int nseq, int L, int heads,
                                          int W, int causal, unsigned* amax, hipStream_t st) {
    constexpr int smem = (64 * (NK64 * 64 + 8) + 64 * ABM_S + 4 * 16 * (NK64 * 64 + 8)) * 2 + 64;
    static bool configured = false;              // benign race (idempotent calls)
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_backward_q_kernel<NK64>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_backward_kv_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, ABK_SMEM) != hipSuccess)
            return CC_ERR_HIP;
        configured = true;
    }
    const int QT = (L + 63) / 64;
    hipLaunchKernelGGL(attention_backward_q_kernel<NK64>, dim3(nseq * heads * QT), dim3(256), smem, st, qkv, d_out, d_qkv, stats, L, heads,
                       W, causal, amax);
    hipLaunchKernelGGL(attention_backward_kv_kernel, dim3(nseq * heads * QT), dim3(256), ABK_SMEM, st, qkv, d_out, d_qkv, stats, L, heads,
                       W, causal, amax);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

extern "C" {

size_t cc_layernorm_backward_workspace_bytes(int32_t rows, int32_t W) {
    if (rows <= 0 || W <= 0) return 0;
    return (size_t)((rows + LNB_ROWS - 1) / LNB_ROWS) * 2 * W * sizeof(float);
}

int cc_layernorm_backward_f32(const float* x, int64_t x_stride, const float* gamma, const float* dy, const float* dres,
                              float* dx, float* dgamma, float* dbeta, int32_t rows, int32_t W, float eps, float* dx_amax,
                              void* ws, size_t ws_bytes, void* stream) {
    if (!x || !gamma || !dy || !dx || !dgamma || !dbeta || rows <= 0 || W <= 0 || (W & 3) || W > 1024) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_layernorm_backward_workspace_bytes(rows, W)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = (rows + LNB_ROWS - 1) / LNB_ROWS;
    hipLaunchKernelGGL(layernorm_backward_kernel, dim3(blocks), dim3(256), 0, st, x, x_stride, gamma, dy, dres, dx,
                       static_cast<float*>(ws), rows, W, eps, reinterpret_cast<unsigned*>(dx_amax));
    hipLaunchKernelGGL(column_reduce_kernel, dim3((2 * W + 31) / 32), dim3(256), 0, st, static_cast<const float*>(ws), blocks, 2 * W,
                       dgamma, dbeta, W);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_quick_gelu_backward_f16(const void* u_pre_f16, const float* du, float* du_pre, int64_t n, float* out_amax, void* stream) {
    if (!u_pre_f16 || !du || !du_pre || n <= 0 || (n & 3)) return CC_ERR_INVALID;
    hipLaunchKernelGGL(quick_gelu_backward_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(u_pre_f16), du, du_pre, n, reinterpret_cast<unsigned*>(out_amax));
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_quick_gelu_f16(const void* in_f16, void* out_f16, int64_t n, void* stream) {
    if (!in_f16 || !out_f16 || n <= 0 || (n & 3)) return CC_ERR_INVALID;
    hipLaunchKernelGGL(quick_gelu_f16_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(in_f16), static_cast<_Float16*>(out_f16), n);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

size_t cc_column_sums_workspace_bytes(int32_t rows, int32_t cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + CS_ROWS - 1) / CS_ROWS) * cols * sizeof(float);
}

int cc_column_sums_f32(const float* in, int32_t rows, int32_t cols, float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_column_sums_workspace_bytes(rows, cols)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int chunks = (rows + CS_ROWS - 1) / CS_ROWS;
    hipLaunchKernelGGL(column_partial_kernel, dim3((cols + 255) / 256, chunks), dim3(256), 0, st, in, rows, cols,
                       static_cast<float*>(ws));
    hipLaunchKernelGGL(column_reduce_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, static_cast<const float*>(ws), chunks,
                       cols, out, out, cols);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

size_t cc_attention_backward_workspace_bytes(int32_t nseq, int32_t L, int32_t heads) {
    if (nseq <= 0 || L <= AB_L || heads <= 0) return 0;         // (short sequences: one launch, no scratch)
    return (size_t)nseq * L * heads * 2 * sizeof(float);
}

int cc_attention_backward_f16(const void* qkv_f16, const float* d_out, float* d_qkv, int32_t nseq, int32_t L, int32_t heads,
                              int32_t W, int32_t causal, float* out_amax, void* ws, size_t ws_bytes, void* stream) {
    if (!qkv_f16 || !d_out || !d_qkv || nseq <= 0 || L <= 0 || heads <= 0 || W != heads * AB_D) return CC_ERR_INVALID;
    if (L > 256) return CC_ERR_UNSUPPORTED;
    if (L > AB_L) {
        if (!ws || ws_bytes < cc_attention_backward_workspace_bytes(nseq, L, heads)) return CC_ERR_WORKSPACE;
        const _Float16* q = static_cast<const _Float16*>(qkv_f16);
        unsigned* am = reinterpret_cast<unsigned*>(out_amax);
        hipStream_t st = static_cast<hipStream_t>(stream);
        switch ((L + 63) / 64) {
            case 2: return launch_attention_backward_long<2>(q, d_out, d_qkv, static_cast<float*>(ws), nseq, L, heads, W, causal, am, st);
            case 3: return launch_attention_backward_long<3>(q, d_out, d_qkv, static_cast<float*>(ws), nseq, L, heads, W, causal, am, st);
            default: return launch_attention_backward_long<4>(q, d_out, d_qkv, static_cast<float*>(ws), nseq, L, heads, W, causal, am, st);
        }
    }
    static bool configured = false;              // benign race (idempotent call)
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_backward_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, AB_SMEM) != hipSuccess)
            return CC_ERR_HIP;
        configured = true;
    }
#ifndef CC_ATTENTION_BACKWARD_FP32
    {
        static bool configured_m = false;        // benign race (idempotent call)
        if (!configured_m) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_backward_mfma_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, ABM_SMEM) != hipSuccess)
                return CC_ERR_HIP;
            configured_m = true;
        }
        hipLaunchKernelGGL(attention_backward_mfma_kernel, dim3(nseq * heads), dim3(256), ABM_SMEM, static_cast<hipStream_t>(stream),
                           static_cast<const _Float16*>(qkv_f16), d_out, d_qkv, L, heads, W, causal,
                           reinterpret_cast<unsigned*>(out_amax));
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
#endif
    hipLaunchKernelGGL(attention_backward_kernel, dim3(nseq * heads), dim3(256), AB_SMEM, static_cast<hipStream_t>(stream),
                       static_cast<const _Float16*>(qkv_f16), d_out, d_qkv, L, heads, W, causal, reinterpret_cast<unsigned*>(out_amax));
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_cast_scaled_f16(const float* in, void* out_f16, int64_t n, float* amax_scratch, float* scale_out, void* stream) {
    if (!in || !out_f16 || !amax_scratch || !scale_out || n <= 0) return CC_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(amax_scratch, 0, sizeof(float), st) != hipSuccess) return CC_ERR_HIP;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n, 256 * 16) < 2048 ? grid_for(n, 256 * 16) : 2048), dim3(256), 0, st, in, n, reinterpret_cast<unsigned*>(amax_scratch));
    hipLaunchKernelGGL(cast_scaled_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, in, static_cast<_Float16*>(out_f16), n,
                       amax_scratch, scale_out);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_unscale_f32(float* x, int64_t n, const float* scale_a, const float* scale_b, void* stream) {
    if (!x || !scale_a || n <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(unscale_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, static_cast<hipStream_t>(stream), x, n, scale_a,
                       scale_b);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

size_t cc_bertadam_workspace_bytes(void) { return BA_BLOCKS * sizeof(double); }

/* One BertAdam step on one parameter tensor (utils/optimization.py:100-170; all tensors fp32, n elements): grad is clipped in
 * place to max_grad_norm (<= 0: no clipping), next_m / next_v updated, param -= lr_scheduled * (m / (sqrt(v) + e) + wd * param).
 * lr_scheduled = lr * schedule(step / t_total, warmup) is the caller's (host arithmetic, centerclip_amd.train.BertAdam);
 * lr_dev != null: read from that device float instead (a step captured into a hipGraph is replayed with new values). */
int cc_bertadam_step_f32(float* param, float* grad, float* next_m, float* next_v, int64_t n, float lr_scheduled, float b1,
                         float b2, float e, float weight_decay, float max_grad_norm, const float* lr_dev, void* ws, size_t ws_bytes,
                         void* stream) {
    if (!param || !grad || !next_m || !next_v || n <= 0) return CC_ERR_INVALID;
    if (!ws || ws_bytes < cc_bertadam_workspace_bytes()) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n <= 8192) {
        hipLaunchKernelGGL(bertadam_small_kernel, dim3(1), dim3(256), 0, st, param, grad, next_m, next_v, (int)n, lr_scheduled, b1, b2, e,
                           weight_decay, max_grad_norm, lr_dev);
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    double* partial = static_cast<double*>(ws);
    const int nb = (int)((n + 1023) / 1024 < BA_BLOCKS ? (n + 1023) / 1024 : BA_BLOCKS);
    if (max_grad_norm > 0.f) hipLaunchKernelGGL(bertadam_norm_kernel, dim3(nb), dim3(256), 0, st, grad, n, partial);
    hipLaunchKernelGGL(bertadam_step_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, st, param, grad, next_m, next_v, n, partial, nb,
                       lr_scheduled, b1, b2, e, weight_decay, max_grad_norm, lr_dev);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

/* BertAdam steps of `count` small tensors (n <= CC_BERTADAM_MULTI_MAX_N each) in one launch: items_dev = count cc_bertadam_item
 * records in device memory (include/centerclip_hip.h); per tensor the arithmetic - and the bits - of cc_bertadam_step_f32. */
int cc_bertadam_multi_f32(const void* items_dev, int32_t count, float b1, float b2, float e, float max_grad_norm, void* stream) {
    if (!items_dev || count <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(bertadam_multi_small_kernel, dim3(count), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const BertAdamItem*>(items_dev), b1, b2, e, max_grad_norm);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

/* The same for `count` LARGE tensors (any n) in TWO launches: items_dev = count cc_bertadam_big_item records in device memory,
 * ordered, with norm_blk0 / step_blk0 = the running sums of norm_blocks / step_blocks = cc_bertadam_norm_blocks(n) /
 * cc_bertadam_step_blocks(n); ws >= total_norm_blocks doubles.  Per tensor the arithmetic and the bits of cc_bertadam_step_f32. */
int32_t cc_bertadam_norm_blocks(int64_t n) { return n <= 0 ? 0 : (int32_t)((n + 1023) / 1024 < BA_BLOCKS ? (n + 1023) / 1024 : BA_BLOCKS); }
int32_t cc_bertadam_step_blocks(int64_t n) { return n <= 0 ? 0 : (int32_t)grid_for(n, 1024); }
int cc_bertadam_multi_large_f32(const void* items_dev, int32_t count, int32_t total_norm_blocks, int32_t total_step_blocks, float b1,
                                float b2, float e, float max_grad_norm, void* ws, size_t ws_bytes, void* stream) {
    if (!items_dev || count <= 0 || total_norm_blocks <= 0 || total_step_blocks <= 0) return CC_ERR_INVALID;
    if (!ws || ws_bytes < (size_t)total_norm_blocks * sizeof(double)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const BertAdamBigItem* items = static_cast<const BertAdamBigItem*>(items_dev);
    double* partial = static_cast<double*>(ws);
    if (max_grad_norm > 0.f)
        hipLaunchKernelGGL(bertadam_multi_norm_kernel, dim3(total_norm_blocks), dim3(256), 0, st, items, count, partial);
    hipLaunchKernelGGL(bertadam_multi_step_kernel, dim3(total_step_blocks), dim3(256), 0, st, items, count, partial, b1, b2, e, max_grad_norm);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

/* fp16 operand copies of a matrix for the backward of a Linear: `in` fp32 [rows, cols] (or in_f16, already fp16) ->
 * out_f16 [rows, cols] (may be null) and out_t_f16 [cols, rows_pad] = its transpose with zero columns behind `rows`
 * (rows_pad >= rows, a multiple of 64; cols % 4 == 0).  scaled != 0: fp32 input scaled by the device-chosen power of two of
 * cc_cast_scaled_f16 (amax_scratch: one device float; *scale_out receives the scale); otherwise scale 1.  col_sums != null
 * (fp32 input only): the column sums of the unscaled matrix [cols] from the same read (a Linear's bias gradient), per 64-row
 * tile partials in ws (cc_cast_transpose_colsum_workspace_bytes) added in tile order. */
size_t cc_cast_transpose_colsum_workspace_bytes(int32_t rows_pad, int32_t cols) {
    return rows_pad > 0 && cols > 0 ? (size_t)(rows_pad / 64) * cols * sizeof(float) : 0;
}
int cc_cast_transpose_f16(const float* in, const void* in_f16, void* out_f16, void* out_t_f16, int32_t rows, int32_t cols,
                          int32_t rows_pad, int32_t scaled, float* amax_scratch, float* scale_out, float* col_sums, void* ws,
                          size_t ws_bytes, void* stream) {
    if ((!in && !in_f16) || (!out_t_f16 && !out_f16) || rows <= 0 || cols <= 0 || (cols & 3) || rows_pad < rows || (rows_pad & 63)) return CC_ERR_INVALID;
    if (scaled && (!in || !amax_scratch || !scale_out)) return CC_ERR_INVALID;
    if (col_sums && (!in || !ws || ws_bytes < cc_cast_transpose_colsum_workspace_bytes(rows_pad, cols))) return col_sums && in ? CC_ERR_WORKSPACE : CC_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (scaled == 1) {                                            // (2: *amax_scratch already holds the largest magnitude)
        if (hipMemsetAsync(amax_scratch, 0, sizeof(float), st) != hipSuccess) return CC_ERR_HIP;
        const unsigned ab = grid_for((int64_t)rows * cols, 256 * 16);
        hipLaunchKernelGGL(absmax_kernel, dim3(ab < 2048 ? ab : 2048), dim3(256), 0, st, in, (int64_t)rows * cols,
                           reinterpret_cast<unsigned*>(amax_scratch));
    }
    hipLaunchKernelGGL(cast_transpose_kernel, dim3((cols + 63) / 64, rows_pad / 64), dim3(256), 0, st, in,
                       static_cast<const _Float16*>(in_f16), static_cast<_Float16*>(out_f16), static_cast<_Float16*>(out_t_f16), rows,
                       cols, rows_pad, scaled ? amax_scratch : nullptr, scaled ? scale_out : nullptr,
                       // (col_sums null with a large enough ws: the per-tile partial column sums only - cc_wgrad_tn_f16 adds them)
                       (in && ws && ws_bytes >= cc_cast_transpose_colsum_workspace_bytes(rows_pad, cols)) ? static_cast<float*>(ws) : nullptr);
    if (col_sums)
        hipLaunchKernelGGL(column_reduce_kernel, dim3((cols + 31) / 32), dim3(256), 0, st, static_cast<const float*>(ws), rows_pad / 64,
                           cols, col_sums, col_sums, cols);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

}  // extern "C"
