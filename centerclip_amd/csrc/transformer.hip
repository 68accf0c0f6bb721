// Non-GEMM kernels of the CLIP forward on gfx950: LayerNorm (fp32 statistics), small-sequence
// multi-head attention on MFMA, patch im2col, embeddings and the projection heads.
// Reference semantics: modules/clip.py:183-189 (LayerNorm in fp32, eps 1e-5), :205,220-226
// (nn.MultiheadAttention: packed in_proj q,k,v; heads = contiguous 64-wide slices; softmax(q k^T /
// sqrt(64) + mask) v), :326-338 (patch embed + CLS + positional embedding + ln_pre), :448-454
// (causal mask), :463-464 / :480-484 (ln_post / ln_final + projection of the CLS / EOT row).
#include "cc_kernels.h"
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ============================================================================ LayerNorm
// One wave per row, row cached in registers (W <= 1024, W % 4 == 0).  Two-pass statistics in
// fp32 (mean, then centred second moment) like ATen.  OUT_F16: write fp16 (GEMM operand) else
// fp32 (may alias the input: ln_pre runs in place).  Rows are addressed as
// in + row*in_stride so a strided subset (CLS rows) can be normalised directly.
struct LnPair {
    LnArgs a[2];
    int blocks0;        // workgroups [0, blocks0) -> a[0]
};

template <bool OUT_F16>
__device__ __forceinline__ void layernorm_row(const LnArgs& a, int row, int lane, float eps) {
    const int W = a.W;
    const float* src = a.in + (int64_t)row * a.in_stride;
    // ln_pre: the CLS row of every frame is class_embedding + positional_embedding[0] (modules/clip.py:334-336), formed
    // here instead of by a pass of its own (cls_period = tokens per frame; the input row is never read)
    const bool cls_row = a.cls && (row % a.cls_period) == 0;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        if (w >= W) v[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (cls_row) {
            const float4 c = *reinterpret_cast<const float4*>(a.cls + w), p0 = *reinterpret_cast<const float4*>(a.pos0 + w);
            v[t] = make_float4(c.x + p0.x, c.y + p0.y, c.z + p0.z, c.w + p0.w);
        } else v[t] = *reinterpret_cast<const float4*>(src + w);
        s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
    const float mean = cc_wave_sum_fast(s) / (float)W;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        if (w < W) {
            const float c0 = v[t].x - mean, c1 = v[t].y - mean, c2 = v[t].z - mean, c3 = v[t].w - mean;
            q += (c0 * c0 + c1 * c1) + (c2 * c2 + c3 * c3);
        }
    }
    const float rstd = 1.0f / sqrtf(cc_wave_sum_fast(q) / (float)W + eps);
    float4 ov[4];
    float ot = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        ov[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w < W) {
            const float4 gm = *reinterpret_cast<const float4*>(a.gamma + w);
            const float4 bt = *reinterpret_cast<const float4*>(a.beta + w);
            const float o0 = (v[t].x - mean) * rstd * gm.x + bt.x, o1 = (v[t].y - mean) * rstd * gm.y + bt.y;
            const float o2 = (v[t].z - mean) * rstd * gm.z + bt.z, o3 = (v[t].w - mean) * rstd * gm.w + bt.w;
            ov[t] = make_float4(o0, o1, o2, o3);
            ot += (o0 + o1) + (o2 + o3);
            if (OUT_F16) {
                h4 o = {(_Float16)o0, (_Float16)o1, (_Float16)o2, (_Float16)o3};
                *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(a.out) + (int64_t)row * a.out_stride + w) = o;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + (int64_t)row * a.out_stride + w) =
                    make_float4(o0, o1, o2, o3);
            }
        }
    }
    if (!OUT_F16 && a.out16) {
        // fp16 copy of the new residual row for the folded LayerNorm of the next block: centred on the row mean (the
        // consumer's LayerNorm is shift invariant; an un-centred copy loses |mean| / sigma in precision), + its
        // (sum, sumsq) and the mean that was subtracted
        const float om = cc_wave_sum_fast(ot) / (float)W;
        float osum = 0.f, osq = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = lane * 4 + t * 256;
            if (w < W) {
                h4 o = {(_Float16)(ov[t].x - om), (_Float16)(ov[t].y - om), (_Float16)(ov[t].z - om), (_Float16)(ov[t].w - om)};
                *reinterpret_cast<h4*>(a.out16 + (int64_t)row * W + w) = o;
                const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
                osum += (q0 + q1) + (q2 + q3);
                osq += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
            }
        }
        if (a.stats) {
            osum = cc_wave_sum_fast(osum);
            osq = cc_wave_sum_fast(osq);
            if (lane == 0) reinterpret_cast<float2*>(a.stats)[row] = make_float2(osum, osq);
        }
        if (a.shift && lane == 0) a.shift[row] = om;
    }
}

template <bool OUT_F16>
__global__ __launch_bounds__(256) void layernorm_kernel(LnPair pr, float eps) {
    const bool second = (int)blockIdx.x >= pr.blocks0;
    const LnArgs& a = second ? pr.a[1] : pr.a[0];
    const int row = ((int)blockIdx.x - (second ? pr.blocks0 : 0)) * 4 + (threadIdx.x >> 6);
    if (row >= a.rows) return;
    layernorm_row<OUT_F16>(a, row, threadIdx.x & 63, eps);
}

// fp16 copy + (sum, sum of squares of the fp16-rounded values) of contiguous fp32 rows; one wave per row (W <= 1024).
// shift != nullptr: the copy is centred on the row mean (written to shift[row]) - see layernorm_row.
__global__ __launch_bounds__(256) void row_stats_kernel(const float* __restrict__ h, _Float16* __restrict__ h16,
                                                        float* __restrict__ stats, float* __restrict__ shift, int rows,
                                                        int W) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float4 v[4];
    float tot = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        v[t] = (w < W) ? *reinterpret_cast<const float4*>(h + (int64_t)row * W + w) : make_float4(0.f, 0.f, 0.f, 0.f);
        tot += (v[t].x + v[t].y) + (v[t].z + v[t].w);
    }
    const float om = shift ? cc_wave_sum_fast(tot) / (float)W : 0.f;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int w = lane * 4 + t * 256;
        if (w < W) {
            h4 o = {(_Float16)(v[t].x - om), (_Float16)(v[t].y - om), (_Float16)(v[t].z - om), (_Float16)(v[t].w - om)};
            *reinterpret_cast<h4*>(h16 + (int64_t)row * W + w) = o;
            const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
            s += (q0 + q1) + (q2 + q3);
            q += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
        }
    }
    s = cc_wave_sum_fast(s);
    q = cc_wave_sum_fast(q);
    if (lane == 0) {
        reinterpret_cast<float2*>(stats)[row] = make_float2(s, q);
        if (shift) shift[row] = om;
    }
}

// LayerNorm folding of a Linear layer: w_out[n,k] = fp16(W[n,k] * gamma[k]); c1[n] = sum_k float(w_out[n,k]);
// c2[n] = sum_k beta[k] * W[n,k] + bias[n].  One workgroup per output row n.
__global__ __launch_bounds__(256) void fold_ln_linear_kernel(const float* __restrict__ Wt, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int N, int K,
                                                             _Float16* __restrict__ w_out, float* __restrict__ c1,
                                                             float* __restrict__ c2) {
    __shared__ float red[8];
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float s1 = 0.f, s2 = 0.f;
    for (int k = tid; k < K; k += 256) {
        const float w = Wt[(int64_t)n * K + k];
        const _Float16 wf = (_Float16)(w * gamma[k]);
        w_out[(int64_t)n * K + k] = wf;
        s1 += (float)wf;
        s2 = fmaf(beta[k], w, s2);
    }
    s1 = cc_wave_sum_fast(s1);
    s2 = cc_wave_sum_fast(s2);
    if (lane == 0) { red[wave] = s1; red[4 + wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        c1[n] = (red[0] + red[1]) + (red[2] + red[3]);
        c2[n] = (red[4] + red[5]) + (red[6] + red[7]) + (bias ? bias[n] : 0.f);
    }
}

// ============================================================================ attention
// One workgroup (4 waves) per (sequence, head); head_dim = 64; L <= 256 tokens.
//   K   -> LDS [KT keys][64] fp16, 128-byte rows, 16-byte chunks XOR-swizzled by (row & 7)
//   V^T -> LDS [64 d][VS halfs]   (VS/2 words == 40 mod 64 -> conflict-free ds_read_b128)
//   per wave, per 16-query tile:  S^T = K Q^T (MFMA, K fragment as A operand so a lane owns 4
//   consecutive keys of one query) -> masked softmax in registers (fp32) -> P (fp16) to a
//   per-wave LDS strip -> O^T = V^T P^T (MFMA) -> fp16 store, 4 consecutive d per lane.
#define ATT_D 64
#define ATT_MAX_KT 256

__host__ __device__ inline int att_vs_halfs(int KT) {      // row stride of V^T / P in halfs
    int words = KT / 2 + 4;
    while ((words & 63) != 40) ++words;
    return words * 2;
}

struct AttPair {
    AttArgs a[2];
    int wgs0;           // workgroups [0, wgs0) -> a[0]
};

__global__ __launch_bounds__(256) void attention_kernel(AttPair pr, float scale) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool second = (int)blockIdx.x >= pr.wgs0;
    const AttArgs at = second ? pr.a[1] : pr.a[0];
    const _Float16* __restrict__ qkv = at.qkv;
    _Float16* __restrict__ out = at.out;
    const int heads = at.heads, W = at.W;
    const bool CAUSAL = at.causal != 0;
    const int wg = (int)blockIdx.x - (second ? pr.wgs0 : 0);
    const int seq = wg / heads, head = wg - seq * heads;
    const int L = at.seq_len ? at.seq_len[seq] : at.L;         // (LDS is sized for at.L, the upper bound)
    const int64_t seq_row0 = at.seq_off ? at.seq_off[seq] : (int64_t)seq * at.seq_rows;
    const int KT = (L + 31) & ~31;
    const int VS = att_vs_halfs(KT);
    _Float16* Ks = reinterpret_cast<_Float16*>(smem);                       // KT * 64
    _Float16* Vt = Ks + KT * ATT_D;                                          // 64 * VS
    _Float16* Ps = Vt + ATT_D * VS;                                          // 4 waves * 16 * VS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t ld = 3 * (int64_t)W * at.tok_rows;          // qkv stride between consecutive tokens of a sequence
    const _Float16* base = qkv + seq_row0 * 3 * W + head * ATT_D;

    // ---- stage K (swizzled) and V^T; rows >= L are zero
    for (int idx = tid; idx < KT * 8; idx += 256) {
        const int r = idx >> 3, c = idx & 7;
        h8 kv = {0, 0, 0, 0, 0, 0, 0, 0}, vv = kv;
        if (r < L) {
            kv = *reinterpret_cast<const h8*>(base + (int64_t)r * ld + W + c * 8);
            vv = *reinterpret_cast<const h8*>(base + (int64_t)r * ld + 2 * W + c * 8);
        }
        *reinterpret_cast<h8*>(reinterpret_cast<unsigned char*>(Ks) + r * 128 + ((c ^ (r & 7)) << 4)) = kv;
#pragma unroll
        for (int e = 0; e < 8; ++e) Vt[(c * 8 + e) * VS + r] = vv[e];
    }
    __syncthreads();

    const int l15 = lane & 15, lg = lane >> 4;
    const int nkt = KT / 16;                       // 16-key tiles
    _Float16* Pw = Ps + wave * 16 * VS;
    const int qtiles = (L + 15) / 16;
    for (int qt = wave; qt < qtiles; qt += 4) {
        const int q = qt * 16 + l15;               // this lane's query row (as B-operand column)
        const int qc = min(q, L - 1);
        h8 qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            qf[ks] = *reinterpret_cast<const h8*>(base + (int64_t)qc * ld + (ks * 4 + lg) * 8);
        // S^T tiles: lane holds S[q][key = kt*16 + lg*4 + r]
        f32x4 s[ATT_MAX_KT / 16];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT / 16; ++kt) {
            if (kt < nkt) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
                const int r = kt * 16 + l15;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const h8 kf = *reinterpret_cast<const h8*>(reinterpret_cast<const unsigned char*>(Ks) + r * 128 +
                                                               (((ks * 4 + lg) ^ (r & 7)) << 4));
                    a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], a, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int key = kt * 16 + lg * 4 + e;
                    const bool ok = key < L && (!CAUSAL || key <= q);
                    a[e] = ok ? a[e] * scale : -3.0e38f;
                    mx = fmaxf(mx, a[e]);
                }
                s[kt] = a;
            }
        }
        mx = cc_rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT / 16; ++kt) {
            if (kt < nkt) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pexp = (s[kt][e] > -1.0e38f) ? __expf(s[kt][e] - mx) : 0.f;
                    s[kt][e] = pexp;
                    sum += pexp;
                }
            }
        }
        sum = cc_rows_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < ATT_MAX_KT / 16; ++kt) {
            if (kt < nkt) {
                h4 ph = {(_Float16)(s[kt][0] * inv), (_Float16)(s[kt][1] * inv), (_Float16)(s[kt][2] * inv),
                         (_Float16)(s[kt][3] * inv)};
                *reinterpret_cast<h4*>(Pw + l15 * VS + kt * 16 + lg * 4) = ph;
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);        // lgkmcnt(0): the strip is private to this wave
        __builtin_amdgcn_wave_barrier();
        // O^T = V^T P^T : lane holds O[q = l15][d = dt*16 + lg*4 + e]
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = 0; kb < KT / 32; ++kb) {
            const h8 pf = *reinterpret_cast<const h8*>(Pw + l15 * VS + kb * 32 + lg * 8);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const h8 vf = *reinterpret_cast<const h8*>(Vt + (dt * 16 + l15) * VS + kb * 32 + lg * 8);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        if (q < L) {
            _Float16* dst = out + (seq_row0 + (int64_t)q * at.tok_rows) * W + head * ATT_D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                h4 oh = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                *reinterpret_cast<h4*>(dst + dt * 16 + lg * 4) = oh;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------- short sequences (L <= 64)
// One WAVE per (sequence, head), four independent waves per workgroup, no workgroup barrier:
//   K fragments are loaded straight from global memory into registers (each is reused by all query tiles
//   of the wave), V is transposed into a per-wave 8 KB LDS tile [64 d][64 keys] whose 16-byte chunks are
//   XOR-swizzled by f(d) = (d ^ d>>3) & 7 - conflict-free both for the 2-byte transposing writes (lanes of a
//   row differ in d>>3) and for the ds_read_b128 operand fetches (16 consecutive d) - and P goes through a
//   per-wave 2 KB strip.  All global loads of a head (7 V + 8 K + 2 Q per lane) are issued up front.
// Covers L = 50 (ViT-B/32 and every K = 49 clustered block) and the 32-token text tower.
#define ATTW_KT 64
// development builds (-DCC_DEV_KNOBS) only: per-workgroup real-time stamps of wave 0 (entry, operands staged, exit), 100 MHz
#ifdef CC_DEV_KNOBS
__device__ long long* g_att_prof = nullptr;
extern "C" void cc_debug_set_att_profile(long long* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_att_prof), &dev_buf, sizeof(dev_buf));
}
#define ATT_PROF_INIT() long long* aprof = g_att_prof
#define ATT_STAMP(slot)                                                                                              \
    do {                                                                                                             \
        if (aprof && threadIdx.x == 0 && blockIdx.x < 4096) aprof[(int64_t)blockIdx.x * 4 + (slot)] = (long long)wall_clock64(); \
    } while (0)
#else
#define ATT_PROF_INIT() do { } while (0)
#define ATT_STAMP(slot) do { } while (0)
#endif
__global__ __launch_bounds__(256) void attention_wave_kernel(AttPair pr, float scale) {
    ATT_PROF_INIT();
    ATT_STAMP(0);
    __shared__ __attribute__((aligned(16))) _Float16 lds[4][ATT_D * ATTW_KT + 2 * 16 * (ATTW_KT + 8)];   // V^T + two P strips
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int unit = blockIdx.x * 4 + wave;
    const int units0 = pr.a[0].nseq * pr.a[0].heads;
    const bool second = unit >= units0;
    const AttArgs at = second ? pr.a[1] : pr.a[0];
    const int u = unit - (second ? units0 : 0);
    if (u >= at.nseq * at.heads) return;                       // wave-uniform; no barriers below
    const int heads = at.heads, W = at.W;
    const bool CAUSAL = at.causal != 0;
    const int seq = u / heads, head = u - seq * heads;
    const int L = at.seq_len ? at.seq_len[seq] : at.L;         // wave-uniform; a compacted caption may be shorter than at.L
    const int64_t seq_row0 = at.seq_off ? at.seq_off[seq] : (int64_t)seq * at.seq_rows;
    const int64_t ld = 3 * (int64_t)W * at.tok_rows;
    const _Float16* base = at.qkv + seq_row0 * 3 * W + head * ATT_D;
    _Float16* Vt = lds[wave];
    _Float16* Pw = Vt + ATT_D * ATTW_KT;
    constexpr int PS = ATTW_KT + 8;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- issue every global load of this head
    h8 vreg[7];
#pragma unroll
    for (int t = 0; t < 7; ++t) {                               // 64 keys x 8 chunks = 512 -> 8 per lane, last is padding
        const int idx = t * 64 + lane, r = idx >> 3, c = idx & 7;
        vreg[t] = (r < L) ? *reinterpret_cast<const h8*>(base + (int64_t)r * ld + 2 * W + c * 8) : h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    h8 kf[4][2];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
        const int r = min(kt * 16 + l15, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = *reinterpret_cast<const h8*>(base + (int64_t)r * ld + W + (ks * 4 + lg) * 8);
    }
    // the query fragments of all (at most 4) query tiles too: behind the loop's own loads every tile would wait a
    // full L2 round trip with fewer than 3 waves per SIMD to cover it
    const int qtiles = (L + 15) / 16;
    h8 qf[4][2];
#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        const int qc = min(qt * 16 + l15, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[qt][ks] = *reinterpret_cast<const h8*>(base + (int64_t)qc * ld + (ks * 4 + lg) * 8);
    }
    // rows 56..63 of V (7 loads cover idx < 448 -> rows < 56): zero them and rows >= L
#pragma unroll
    for (int t = 0; t < 7; ++t) {
        const int idx = t * 64 + lane, r = idx >> 3, c = idx & 7;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = c * 8 + e;
            Vt[d * ATTW_KT + ((((r >> 3) ^ ((d ^ (d >> 3)) & 7))) << 3) + (r & 7)] = vreg[t][e];
        }
    }
    {   // keys 56..63 are beyond every supported L (<= 56 rows loaded): zero fill
        const int d = lane;
        *reinterpret_cast<h8*>(Vt + d * ATTW_KT + ((7 ^ ((d ^ (d >> 3)) & 7)) << 3)) = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    ATT_STAMP(1);

#pragma unroll
    for (int qt = 0; qt < 4; ++qt) {
        if (qt >= qtiles) break;                              // wave-uniform
        const int q = qt * 16 + l15;
        _Float16* Pq = Pw + (qt & 1) * (16 * PS);             // alternate strips: tile qt+1 need not wait for tile qt's reads
        f32x4 s[4];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][ks], qf[qt][ks], a, 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = kt * 16 + lg * 4 + e;
                const bool ok = key < L && (!CAUSAL || key <= q);
                a[e] = ok ? a[e] * scale : -3.0e38f;
                mx = fmaxf(mx, a[e]);
            }
            s[kt] = a;
        }
        mx = cc_rows_max(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pexp = (s[kt][e] > -1.0e38f) ? __expf(s[kt][e] - mx) : 0.f;
                s[kt][e] = pexp;
                sum += pexp;
            }
        sum = cc_rows_sum(sum);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            // two roundings (fp32 product, then fp16), pinned: the in_proj + attention form of gemm.hip computes the same bits
            float p0 = s[kt][0] * inv, p1 = s[kt][1] * inv, p2 = s[kt][2] * inv, p3 = s[kt][3] * inv;
            asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            h4 ph = {(_Float16)p0, (_Float16)p1, (_Float16)p2, (_Float16)p3};
            *reinterpret_cast<h4*>(Pq + l15 * PS + kt * 16 + lg * 4) = ph;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        f32x4 o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const h8 pf = *reinterpret_cast<const h8*>(Pq + l15 * PS + kb * 32 + lg * 8);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int d = dt * 16 + l15;
                const h8 vf = *reinterpret_cast<const h8*>(Vt + d * ATTW_KT + (((kb * 4 + lg) ^ ((d ^ (d >> 3)) & 7)) << 3));
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
            }
        }
        if (q < L) {
            _Float16* dst = at.out + (seq_row0 + (int64_t)q * at.tok_rows) * W + head * ATT_D;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                h4 oh = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                *reinterpret_cast<h4*>(dst + dt * 16 + lg * 4) = oh;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    ATT_STAMP(2);
}
#undef ATT_STAMP

// ============================================================================ embeddings
// conv1 as GEMM: A[f*n + (ph*g + pw)][c*p*p + kh*p + kw] = video[f][c][ph*p+kh][pw*p+kw]  (fp16)
__global__ __launch_bounds__(256) void im2col_f16_kernel(const float* __restrict__ video, _Float16* __restrict__ A,
                                                         int F, int res, int p) {
    const int g = res / p, n = g * g, Kc = 3 * p * p;
    const int64_t total = (int64_t)F * n * Kc / 8;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t e = idx * 8;
        const int k = (int)(e % Kc);
        const int64_t row = e / Kc;
        const int f = (int)(row / n), pi = (int)(row - (int64_t)f * n);
        const int ph = pi / g, pw = pi - ph * g;
        const int c = k / (p * p), rem = k - c * p * p, kh = rem / p, kw = rem - kh * p;
        const float* src = video + (((int64_t)f * 3 + c) * res + (ph * p + kh)) * res + pw * p + kw;
        const float4 a = *reinterpret_cast<const float4*>(src);
        const float4 b = *reinterpret_cast<const float4*>(src + 4);
        h8 o = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w,
                (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        *reinterpret_cast<h8*>(A + e) = o;
    }
}

// linear_patch = '3d' (modules/clip.py:296-317, Conv3d kernel (3, p, p), stride (1, p, p), padding (1, 0, 0)) as GEMM:
// A[f*n + pi][c*3*p*p + kt*p*p + kh*p + kw] = video[f + kt - 1][c][ph*p+kh][pw*p+kw] when frame f + kt - 1 belongs to the same
// clip of T frames, else 0 (the zero padding along t).
__global__ __launch_bounds__(256) void im2col3d_f16_kernel(const float* __restrict__ video, _Float16* __restrict__ A,
                                                           int F, int T, int res, int p) {
    const int g = res / p, n = g * g, pp = p * p, Kc = 9 * pp;
    const int64_t total = (int64_t)F * n * Kc / 8;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t e = idx * 8;
        const int k = (int)(e % Kc);
        const int64_t row = e / Kc;
        const int f = (int)(row / n), pi = (int)(row - (int64_t)f * n);
        const int ph = pi / g, pw = pi - ph * g;
        const int c = k / (3 * pp), kt = (k / pp) % 3, rem = k % pp, kh = rem / p, kw = rem - kh * p;
        const int ft = (f % T) + kt - 1;
        h8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ft >= 0 && ft < T) {
            const float* src = video + (((int64_t)(f + kt - 1) * 3 + c) * res + (ph * p + kh)) * res + pw * p + kw;
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            o = h8{(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w,
                   (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
        }
        *reinterpret_cast<h8*>(A + e) = o;
    }
}

// N3: the same patch gather from uint8 frames, with the loader's normalisation fused in.  Per sample exactly the
// reference's three fp32 operations, in its order: u/255 (transforms.py:166), - mean, / std (torchvision normalize),
// all IEEE (no reciprocal, no contraction possible between them), so the fp16 patch matrix is bit-identical to the
// one the fp32 path builds from the loader's output.  A byte has 256 values: every workgroup evaluates the 3 x 256
// results once into an LDS table (two IEEE divisions per sample would make the kernel VALU-bound: 35 us vs 29 us
// for the fp32 gather) and the gather itself is a table look-up.  HWC = the decoder's layout: a lane reads 8 pixels
// = 24 contiguous bytes and writes one 16-byte group into each of the 3 channel planes of the im2col row.
template <bool HWC>
__global__ __launch_bounds__(256) void im2col_u8_f16_kernel(const unsigned char* __restrict__ video,
                                                            _Float16* __restrict__ A, int F, int res, int p,
                                                            float m0, float m1, float m2, float s0, float s1, float s2) {
    __shared__ _Float16 table[3][256];
    {
        const int u = threadIdx.x;                                   // blockDim.x == 256
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            float v = (float)u / 255.0f;
            v = v - mean;
            v = v / sd;
            table[c][u] = (_Float16)v;
        }
    }
    __syncthreads();
    const int g = res / p, n = g * g, Kc = 3 * p * p, pp = p * p;
    if (HWC) {
        const int64_t total = (int64_t)F * n * pp / 8;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int64_t e = idx * 8;
            const int rem = (int)(e % pp);
            const int64_t row = e / pp;
            const int f = (int)(row / n), pi = (int)(row - (int64_t)f * n);
            const int ph = pi / g, pw = pi - ph * g;
            const int kh = rem / p, kw = rem - kh * p;
            const unsigned char* src = video + (((int64_t)f * res + ph * p + kh) * res + pw * p + kw) * 3;   // 8-byte aligned
            const uint2 w0 = *reinterpret_cast<const uint2*>(src);
            const uint2 w1 = *reinterpret_cast<const uint2*>(src + 8);
            const uint2 w2 = *reinterpret_cast<const uint2*>(src + 16);
            const unsigned wd[6] = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y};
            _Float16* dst = A + row * Kc + rem;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                h8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int byte = 3 * q + c;
                    o[q] = table[c][(wd[byte >> 2] >> (8 * (byte & 3))) & 0xFFu];
                }
                *reinterpret_cast<h8*>(dst + c * pp) = o;
            }
        }
    } else {
        const int64_t total = (int64_t)F * n * Kc / 8;
        for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
            const int64_t e = idx * 8;
            const int k = (int)(e % Kc);
            const int64_t row = e / Kc;
            const int f = (int)(row / n), pi = (int)(row - (int64_t)f * n);
            const int ph = pi / g, pw = pi - ph * g;
            const int c = k / pp, rem = k - c * pp, kh = rem / p, kw = rem - kh * p;
            const unsigned char* src = video + (((int64_t)f * 3 + c) * res + ph * p + kh) * res + pw * p + kw;
            const uint2 w = *reinterpret_cast<const uint2*>(src);              // x is a multiple of 8, res of 8
            h8 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[q] = table[c][(w.x >> (8 * q)) & 0xFFu];
                o[4 + q] = table[c][(w.y >> (8 * q)) & 0xFFu];
            }
            *reinterpret_cast<h8*>(A + e) = o;
        }
    }
}

// The same gather with the index arithmetic cut down to shifts (p a power of two: every CLIP patch size): the kernel above
// spends ~300 instructions per 16 output bytes on two 64-bit divisions and five 32-bit ones - with 1 B/px coming in it was
// instruction-bound, not HBM-bound (29 us, the time of the fp32 gather that reads 4x the bytes).  One workgroup per strip
// (frame f, patch row ph): the g patches of the strip are g consecutive rows of A, so consecutive items write consecutive
// 16-byte groups; one scalar division per workgroup, one constant division (by 3 or 9) per item.
// P3D: linear_patch = '3d' from uint8 frames (columns (c, kt, kh, kw), frame f + kt - 1 of the same clip or zeros).
template <bool HWC, bool P3D>
__global__ __launch_bounds__(256) void im2col_u8_strip_kernel(const unsigned char* __restrict__ video,
                                                              _Float16* __restrict__ A, int F, int T, int res, int lp,
                                                              float m0, float m1, float m2, float s0, float s1, float s2) {
    __shared__ _Float16 table[3][256];
    {
        const int u = threadIdx.x;                                   // blockDim.x == 256
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
            float v = (float)u / 255.0f;
            v = v - mean;
            v = v / sd;
            table[c][u] = (_Float16)v;
        }
    }
    __syncthreads();
    constexpr int KT = P3D ? 3 : 1;
    const int g = res >> lp, n = g * g, lpp8 = 2 * lp - 3, lp8 = lp - 3, pp = 1 << (2 * lp);
    const int Kc = 3 * KT * pp;
    const int f = (int)blockIdx.x / g, ph = (int)blockIdx.x - f * g;
    const int fc = P3D ? f % T : 0;                                  // position of the frame inside its clip
    _Float16* Arow = A + ((int64_t)f * n + (int64_t)ph * g) * Kc;
    const int rmask = (1 << lpp8) - 1, wmask = (1 << lp8) - 1;
    const h8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
    if (HWC) {
        const int items = g * KT << lpp8;                            // order [pw][kt][kh][kw / 8]
        for (int idx = threadIdx.x; idx < items; idx += 256) {
            const int hi = idx >> lpp8, rem = idx & rmask;
            const int pw = hi / KT, kt = hi - pw * KT;
            const int kh = rem >> lp8, kw = (rem & wmask) << 3;
            _Float16* dst = Arow + (int64_t)pw * Kc + (kt << (2 * lp)) + (rem << 3);
            const bool live = !P3D || (fc + kt - 1 >= 0 && fc + kt - 1 < T);
            unsigned wd[6] = {0, 0, 0, 0, 0, 0};
            if (live) {
                const unsigned char* src = video + (((int64_t)(f + (P3D ? kt - 1 : 0)) * res + (ph << lp) + kh) * res + (pw << lp) + kw) * 3;
                const uint2 w0 = *reinterpret_cast<const uint2*>(src);
                const uint2 w1 = *reinterpret_cast<const uint2*>(src + 8);
                const uint2 w2 = *reinterpret_cast<const uint2*>(src + 16);
                wd[0] = w0.x; wd[1] = w0.y; wd[2] = w1.x; wd[3] = w1.y; wd[4] = w2.x; wd[5] = w2.y;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                h8 o;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int byte = 3 * q + c;
                    o[q] = table[c][(wd[byte >> 2] >> (8 * (byte & 3))) & 0xFFu];
                }
                *reinterpret_cast<h8*>(dst + c * KT * pp) = live ? o : zero;
            }
        }
    } else {
        const int items = g * 3 * KT << lpp8;                        // order [pw][c][kt][kh][kw / 8] = the columns of A
        for (int idx = threadIdx.x; idx < items; idx += 256) {
            const int hi = idx >> lpp8, rem = idx & rmask;
            const int pw = hi / (3 * KT), q = hi - pw * (3 * KT);
            const int c = q / KT, kt = q - c * KT;
            const int kh = rem >> lp8, kw = (rem & wmask) << 3;
            _Float16* dst = Arow + (int64_t)pw * Kc + (q << (2 * lp)) + (rem << 3);
            h8 o = zero;
            if (!P3D || (fc + kt - 1 >= 0 && fc + kt - 1 < T)) {
                const unsigned char* src = video + (((int64_t)(f + (P3D ? kt - 1 : 0)) * 3 + c) * res + (ph << lp) + kh) * res + (pw << lp) + kw;
                const uint2 w = *reinterpret_cast<const uint2*>(src);          // x is a multiple of 8, res of 8
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    o[u] = table[c][(w.x >> (8 * u)) & 0xFFu];
                    o[4 + u] = table[c][(w.y >> (8 * u)) & 0xFFu];
                }
            }
            *reinterpret_cast<h8*>(dst) = o;
        }
    }
}

// text: h[b*Lt + t] = token_embedding[ids[b,t]] + positional_embedding[t]; eot[b] = first argmax ids[b,:]
// (optionally also the fp16 copy of the row and its (sum, sum of squares), exactly as row_stats_kernel forms them)
// first position of the largest id of caption b (the EOT token, modules/clip.py:484), one wave
__device__ __forceinline__ int text_eot_position(const long long* ids, int b, int Lt, int lane) {
    unsigned long long key = 0ull;
    for (int u = lane; u < Lt; u += 64) {
        const unsigned long long k2 = ((unsigned long long)(ids[(int64_t)b * Lt + u] + 0x40000000LL) << 32) |
                                      (unsigned)(0xFFFFFFFFu - (unsigned)u);
        key = k2 > key ? k2 : key;
    }
    key = cc_wave_max_u64(key);
    return (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
}

__device__ __forceinline__ void text_embed_row(const TextEmbedArgs& e, int row, int lane) {
    const long long* ids = e.ids;
    const int Lt = e.Lt, W = e.W;
    int* eot = e.eot;
    const int b = row / Lt, t = row - b * Lt;
    const int src_row = row;                                     // (b, t) in the id grid
    if (e.seq_off) {
        // compaction: this wave works out where caption b starts (the EOT positions of the captions before it: Bt is a
        // batch of captions, a few wave reductions over L2-resident ids) and drops the row if it lies behind the EOT
        // (lane l scans caption base + l on its own - Lt independent loads - and one wave sum per 64 captions adds the
        // lengths: a chain of one scan + one reduction instead of one wave reduction per earlier caption, which made the last
        // caption's rows the long pole of the whole pre-stage launch: up to 16 dependent reductions)
        int off = 0, my_eot = 0;
        for (int base = 0; base <= b; base += 64) {
            const int bb = base + lane;
            unsigned long long key = 0ull;
            if (bb <= b) {
                const long long* row_ids = ids + (int64_t)bb * Lt;
                for (int u = 0; u < Lt; ++u) {
                    const unsigned long long k2 = ((unsigned long long)(row_ids[u] + 0x40000000LL) << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)u);
                    key = k2 > key ? k2 : key;
                }
            }
            const int pos = (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull));
            int len = bb < b ? pos + 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) len += __shfl_xor(len, o, 64);
            off += len;
            if (b - base < 64) my_eot = __shfl(pos, b - base, 64);
        }
        if (t == 0 && lane == 0) {
            e.seq_off[b] = off;
            e.seq_len[b] = my_eot + 1;
            eot[b] = off + my_eot;                                // absolute row of the EOT token
            if (b == e.Bt - 1) *e.m_total = off + my_eot + 1;
        }
        if (t > my_eot) return;
        row = off + t;
    }
    long long id = ids[src_row];
    if (e.vocab > 0) id = id < 0 ? 0 : (id >= e.vocab ? e.vocab - 1 : id);
    const float* src = e.tok_emb + (int64_t)id * W;
    float4 v[4];
    float tot = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int w = lane * 4 + u * 256;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (w < W) {
            const float4 a = *reinterpret_cast<const float4*>(src + w);
            const float4 pe = *reinterpret_cast<const float4*>(e.pos + (int64_t)t * W + w);
            v[u] = make_float4(a.x + pe.x, a.y + pe.y, a.z + pe.z, a.w + pe.w);
            *reinterpret_cast<float4*>(e.h + (int64_t)row * W + w) = v[u];
            tot += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        }
    }
    if (e.h16) {                                   // centred fp16 copy + its statistics + the mean (see layernorm_row)
        const float om = cc_wave_sum_fast(tot) / (float)W;
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int w = lane * 4 + u * 256;
            if (w < W) {
                h4 o = {(_Float16)(v[u].x - om), (_Float16)(v[u].y - om), (_Float16)(v[u].z - om), (_Float16)(v[u].w - om)};
                *reinterpret_cast<h4*>(e.h16 + (int64_t)row * W + w) = o;
                const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
                s += (q0 + q1) + (q2 + q3);
                q += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
            }
        }
        s = cc_wave_sum_fast(s);
        q = cc_wave_sum_fast(q);
        if (lane == 0) {
            reinterpret_cast<float2*>(e.stats)[row] = make_float2(s, q);
            if (e.shift) e.shift[row] = om;
        }
    }
    if (t == 0 && !e.seq_off) {      // one wave scans the caption for the first maximum id (modules/clip.py:484)
        const int p = text_eot_position(ids, b, Lt, lane);
        if (lane == 0) eot[b] = p;
    }
}

__global__ __launch_bounds__(256) void text_embed_kernel(TextEmbedArgs e) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < e.Bt * e.Lt) text_embed_row(e, row, threadIdx.x & 63);
}

// The stage in front of the first blocks of both towers in ONE launch: ln_pre of the visual rows (in place, CLS rows
// formed on the fly, + fp16 copy + row statistics) in the last blocks_ln workgroups, the text embedding rows in front of them.
__global__ __launch_bounds__(256) void pre_stage_kernel(LnArgs ln, TextEmbedArgs te, int blocks_ln, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the text rows (a longer dependent chain each: EOT search over the earlier captions, table gather) are dispatched first
    const int blocks_te = (int)gridDim.x - blocks_ln;
    if ((int)blockIdx.x >= blocks_te) {
        const int row = ((int)blockIdx.x - blocks_te) * 4 + wave;
        if (row < ln.rows) layernorm_row<false>(ln, row, lane, eps);
    } else {
        const int row = (int)blockIdx.x * 4 + wave;
        if (row < te.Bt * te.Lt) text_embed_row(te, row, lane);
    }
}

// out[r][e0:e0+64] = LN(h[row_of(r)]) @ proj[W, E]   (fp32 throughout).  grid = (E/64, R): every
// workgroup re-normalises its row (W floats, cheap) and produces 64 outputs; the 256 threads are
// 16 k-slices x 16 lanes x 4 outputs, proj reads are 16-byte loads coalesced over the output index.
//   row_of(r) = r*row_mul + (row_idx ? row_idx[r] : 0)
// Up to two heads per launch (rows [0, p[0].R) of the grid's y axis -> p[0], the rest -> p[1]): the visual CLS rows
// and the text EOT rows of the fused forward.
struct HeadPair {
    HeadArgs p[2];
};
__global__ __launch_bounds__(256) void head_project_kernel(HeadPair hp, float eps) {
    __shared__ float xn[1024];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) float part[16][64];
    const bool second = (int)blockIdx.y >= hp.p[0].R;
    const HeadArgs& a = second ? hp.p[1] : hp.p[0];
    const float* __restrict__ h = a.h;
    const int* __restrict__ row_idx = a.row_idx;
    const float* __restrict__ gamma = a.gamma;
    const float* __restrict__ beta = a.beta;
    const float* __restrict__ proj = a.proj;
    float* __restrict__ out = a.out;
    const int row_mul = a.row_mul, W = a.W, E = a.E;
    if ((int)blockIdx.x * 64 >= E) return;                    // the two heads may differ in E (grid x covers the larger)
    const int r = (int)blockIdx.y - (second ? hp.p[0].R : 0), tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* src = h + ((int64_t)r * row_mul + (row_idx ? row_idx[r] : 0)) * W;
    float s = 0.f;
    for (int w = tid; w < W; w += 256) { const float v = src[w]; xn[w] = v; s += v; }
    s = cc_wave_sum_fast(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)W;
    float q = 0.f;
    for (int w = tid; w < W; w += 256) { const float d = xn[w] - mean; q += d * d; }
    q = cc_wave_sum_fast(q);
    if (lane == 0) red[4 + wave] = q;
    __syncthreads();
    const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)W + eps);
    for (int w = tid; w < W; w += 256) xn[w] = (xn[w] - mean) * rstd * gamma[w] + beta[w];
    __syncthreads();
    // 16 k-slices (4 waves x 4 lane groups) x 16 lanes x 4 outputs: 16-byte proj loads, 8 in flight per lane, so the
    // dependent-load chain is W / 16 / 8 batches long (it was W / 4 / 8 with one output per lane)
    const int eg = lane & 15, ks = wave * 4 + (lane >> 4);
    const int e = blockIdx.x * 64 + eg * 4;
    const int wq = (W + 15) / 16, w0 = ks * wq, w1 = min(W, w0 + wq);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < E) {                                              // E % 4 == 0 (checked by the launcher)
        const float* pp = proj + e;
        int w = w0;
        for (; w + 8 <= w1; w += 8) {
            float4 pv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) pv[u] = *reinterpret_cast<const float4*>(pp + (int64_t)(w + u) * E);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float x = xn[w + u];
                acc.x = fmaf(x, pv[u].x, acc.x); acc.y = fmaf(x, pv[u].y, acc.y);
                acc.z = fmaf(x, pv[u].z, acc.z); acc.w = fmaf(x, pv[u].w, acc.w);
            }
        }
        for (; w < w1; ++w) {
            const float4 pv = *reinterpret_cast<const float4*>(pp + (int64_t)w * E);
            const float x = xn[w];
            acc.x = fmaf(x, pv.x, acc.x); acc.y = fmaf(x, pv.y, acc.y); acc.z = fmaf(x, pv.z, acc.z); acc.w = fmaf(x, pv.w, acc.w);
        }
    }
    *reinterpret_cast<float4*>(&part[ks][eg * 4]) = acc;
    __syncthreads();
    if (tid < 64 && (int)blockIdx.x * 64 + tid < E) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][tid];       // fixed order: deterministic
        out[(int64_t)r * E + (int)blockIdx.x * 64 + tid] = t;
    }
}

// ============================================================================ C ABI (single ops)
extern "C" {

int cc_fold_layernorm_linear_f32(const float* weight, const float* bias, const float* gamma, const float* beta,
                                 int32_t N, int32_t K, void* w_f16_out, float* c1_out, float* c2_out, void* stream) {
    if (!weight || !gamma || !beta || !w_f16_out || !c1_out || !c2_out || N <= 0 || K <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(fold_ln_linear_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), weight, bias, gamma,
                       beta, N, K, static_cast<_Float16*>(w_f16_out), c1_out, c2_out);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_row_stats_f16(const float* h, void* h16_out, float* stats_out, float* shift_out, int32_t rows, int32_t W,
                     void* stream) {
    if (!h || !h16_out || !stats_out || rows <= 0 || W <= 0) return CC_ERR_INVALID;
    return cc_launch_row_stats(h, static_cast<_Float16*>(h16_out), stats_out, shift_out, rows, W,
                               static_cast<hipStream_t>(stream));
}

int cc_layernorm_f32(const float* in, int64_t in_stride, const float* gamma, const float* beta, void* out,
                     int64_t out_stride, int32_t rows, int32_t W, float eps, int32_t out_f16, void* stream) {
    if (!in || !gamma || !beta || !out || rows <= 0 || W <= 0 || (W & 3) || W > 1024) return CC_ERR_INVALID;
    LnArgs a{};
    a.in = in; a.in_stride = in_stride; a.gamma = gamma; a.beta = beta; a.out = out; a.out_stride = out_stride;
    a.rows = rows; a.W = W;
    return cc_launch_layernorm2(a, nullptr, eps, out_f16, static_cast<hipStream_t>(stream));
}

int cc_attention_f16(const void* qkv_f16, void* out_f16, int32_t nseq, int32_t L, int32_t heads, int32_t W,
                     int32_t causal, void* stream) {
    if (!qkv_f16 || !out_f16 || nseq <= 0 || L <= 0 || heads <= 0 || W != heads * ATT_D) return CC_ERR_INVALID;
    if (L > ATT_MAX_KT) return CC_ERR_UNSUPPORTED;
    AttArgs a{static_cast<const _Float16*>(qkv_f16), static_cast<_Float16*>(out_f16), nseq, L, heads, W, causal, 0, 0};
    return cc_launch_attention2(a, nullptr, static_cast<hipStream_t>(stream));
}

int cc_attention_strided_f16(const void* qkv_f16, void* out_f16, int32_t nseq, int32_t L, int32_t heads, int32_t W,
                             int32_t causal, int64_t seq_rows, int64_t tok_rows, void* stream) {
    if (!qkv_f16 || !out_f16 || nseq <= 0 || L <= 0 || heads <= 0 || W != heads * ATT_D) return CC_ERR_INVALID;
    if (seq_rows <= 0 || tok_rows <= 0) return CC_ERR_INVALID;
    if (L > ATT_MAX_KT) return CC_ERR_UNSUPPORTED;
    AttArgs a{static_cast<const _Float16*>(qkv_f16), static_cast<_Float16*>(out_f16), nseq, L, heads, W, causal, seq_rows,
              tok_rows};
    return cc_launch_attention2(a, nullptr, static_cast<hipStream_t>(stream));
}

}  // extern "C"

// ---- launch helpers used by clip_forward.hip (same translation-unit-external linkage)
int cc_launch_layernorm2(const LnArgs& a0, const LnArgs* a1, float eps, int out_f16, hipStream_t st) {
    LnPair pr{};
    pr.a[0] = a0;
    pr.a[1] = a1 ? *a1 : a0;
    pr.blocks0 = (a0.rows + 3) / 4;
    const int total = pr.blocks0 + (a1 ? (a1->rows + 3) / 4 : 0);
    if (a0.W > 1024 || (a0.W & 3) || (a1 && (a1->W > 1024 || (a1->W & 3)))) return CC_ERR_INVALID;
    if (out_f16)
        hipLaunchKernelGGL(layernorm_kernel<true>, dim3(total), dim3(256), 0, st, pr, eps);
    else
        hipLaunchKernelGGL(layernorm_kernel<false>, dim3(total), dim3(256), 0, st, pr, eps);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// the same for rows wider than the register-resident kernel holds (W > 1024): the row is streamed twice
__global__ __launch_bounds__(256) void row_stats_wide_kernel(const float* __restrict__ h, _Float16* __restrict__ h16,
                                                             float* __restrict__ stats, float* __restrict__ shift,
                                                             int rows, int W) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float om = 0.f;
    if (shift) {
        float tot = 0.f;
        for (int w = lane * 4; w < W; w += 256) {
            const float4 v = *reinterpret_cast<const float4*>(h + (int64_t)row * W + w);
            tot += (v.x + v.y) + (v.z + v.w);
        }
        om = cc_wave_sum_fast(tot) / (float)W;
    }
    float s = 0.f, q = 0.f;
    for (int w = lane * 4; w < W; w += 256) {
        const float4 v = *reinterpret_cast<const float4*>(h + (int64_t)row * W + w);
        h4 o = {(_Float16)(v.x - om), (_Float16)(v.y - om), (_Float16)(v.z - om), (_Float16)(v.w - om)};
        *reinterpret_cast<h4*>(h16 + (int64_t)row * W + w) = o;
        const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
        s += (q0 + q1) + (q2 + q3);
        q += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
    }
    s = cc_wave_sum_fast(s);
    q = cc_wave_sum_fast(q);
    if (lane == 0) {
        reinterpret_cast<float2*>(stats)[row] = make_float2(s, q);
        if (shift) shift[row] = om;
    }
}

int cc_launch_row_stats(const float* h, _Float16* h16, float* stats, float* shift, int rows, int W, hipStream_t st) {
    if (W & 3) return CC_ERR_INVALID;
    if (W > 1024)
        hipLaunchKernelGGL(row_stats_wide_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, h, h16, stats, shift, rows, W);
    else
        hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, h, h16, stats, shift, rows, W);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

static size_t att_smem_bytes(int L) {
    const int KT = (L + 31) & ~31, VS = att_vs_halfs(KT);
    return (size_t)(KT * ATT_D + ATT_D * VS + 4 * 16 * VS) * 2;
}

int cc_launch_attention2(const AttArgs& a0, const AttArgs* a1, hipStream_t st) {
    if (a0.L > ATT_MAX_KT || (a1 && a1->L > ATT_MAX_KT)) return CC_ERR_UNSUPPORTED;
    AttPair pr{};
    pr.a[0] = a0;
    pr.a[1] = a1 ? *a1 : a0;
    for (int q = 0; q < 2; ++q)
        if (pr.a[q].seq_rows == 0 && pr.a[q].tok_rows == 0) { pr.a[q].seq_rows = pr.a[q].L; pr.a[q].tok_rows = 1; }
    pr.wgs0 = a0.nseq * a0.heads;
    const int total = pr.wgs0 + (a1 ? a1->nseq * a1->heads : 0);
#ifdef CC_DEV_KNOBS
    static const bool wave_path = !(getenv("CC_ATT_BLOCK") && getenv("CC_ATT_BLOCK")[0] == '1');   // A/B switch
#else
    constexpr bool wave_path = true;
#endif
    if (wave_path && a0.L <= 56 && (!a1 || a1->L <= 56)) {          // one wave per (sequence, head)
        hipLaunchKernelGGL(attention_wave_kernel, dim3((total + 3) / 4), dim3(256), 0, st, pr, 0.125f);
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    size_t smem = att_smem_bytes(a0.L);
    if (a1 && att_smem_bytes(a1->L) > smem) smem = att_smem_bytes(a1->L);
    static size_t configured = 64 * 1024;
    if (smem > configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem) != hipSuccess)
            return CC_ERR_HIP;
        configured = smem;
    }
    hipLaunchKernelGGL(attention_kernel, dim3(total), dim3(256), smem, st, pr, 0.125f);
    CC_LAUNCH_CHECK();
    return CC_OK;
}


int cc_launch_im2col3d(const cc_frames& fr, _Float16* A, int F, int T, int res, int p, hipStream_t st) {
    if ((p & 7) || res % p || !fr.data || T <= 0 || F % T) return CC_ERR_INVALID;
    if (fr.format == CC_FRAMES_U8_CHW || fr.format == CC_FRAMES_U8_HWC) {   // uint8 frames: the strip kernel (p a power of two)
        if ((res & 7) || (p & (p - 1))) return (res & 7) ? CC_ERR_INVALID : CC_ERR_UNSUPPORTED;
        for (int c = 0; c < 3; ++c)
            if (!(fr.std[c] > 0.f)) return CC_ERR_INVALID;
        const unsigned char* v = static_cast<const unsigned char*>(fr.data);
        const int lp = __builtin_ctz((unsigned)p);
        if (fr.format == CC_FRAMES_U8_HWC)
            hipLaunchKernelGGL((im2col_u8_strip_kernel<true, true>), dim3(F * (res / p)), dim3(256), 0, st, v, A, F, T, res, lp,
                               fr.mean[0], fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
        else
            hipLaunchKernelGGL((im2col_u8_strip_kernel<false, true>), dim3(F * (res / p)), dim3(256), 0, st, v, A, F, T, res, lp,
                               fr.mean[0], fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    if (fr.format != CC_FRAMES_F32_CHW) return CC_ERR_INVALID;
    const int64_t total = (int64_t)F * (res / p) * (res / p) * 9 * p * p / 8;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(im2col3d_f16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(fr.data), A, F, T, res, p);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_launch_im2col(const cc_frames& fr, _Float16* A, int F, int res, int p, hipStream_t st) {
    if ((p & 7) || res % p || !fr.data) return CC_ERR_INVALID;
    const int64_t total = (int64_t)F * (res / p) * (res / p) * 3 * p * p / 8;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (fr.format == CC_FRAMES_F32_CHW) {
        hipLaunchKernelGGL(im2col_f16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<const float*>(fr.data), A, F, res, p);
    } else if (fr.format == CC_FRAMES_U8_CHW || fr.format == CC_FRAMES_U8_HWC) {
        if (res & 7) return CC_ERR_INVALID;
        for (int c = 0; c < 3; ++c)
            if (!(fr.std[c] > 0.f)) return CC_ERR_INVALID;
        const unsigned char* v = static_cast<const unsigned char*>(fr.data);
        if (!(p & (p - 1))) {                                              // every CLIP patch size: the strip kernel
            const int lp = __builtin_ctz((unsigned)p);
            if (fr.format == CC_FRAMES_U8_HWC)
                hipLaunchKernelGGL((im2col_u8_strip_kernel<true, false>), dim3(F * (res / p)), dim3(256), 0, st, v, A, F, 1, res, lp,
                                   fr.mean[0], fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
            else
                hipLaunchKernelGGL((im2col_u8_strip_kernel<false, false>), dim3(F * (res / p)), dim3(256), 0, st, v, A, F, 1, res, lp,
                                   fr.mean[0], fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
        } else if (fr.format == CC_FRAMES_U8_HWC)
            hipLaunchKernelGGL(im2col_u8_f16_kernel<true>, dim3(blocks), dim3(256), 0, st, v, A, F, res, p, fr.mean[0],
                               fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
        else
            hipLaunchKernelGGL(im2col_u8_f16_kernel<false>, dim3(blocks), dim3(256), 0, st, v, A, F, res, p, fr.mean[0],
                               fr.mean[1], fr.mean[2], fr.std[0], fr.std[1], fr.std[2]);
    } else {
        return CC_ERR_INVALID;
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_launch_text_embed(const TextEmbedArgs& e, hipStream_t st) {
    if ((e.W & 3) || e.W > 1024 || (e.h16 && !e.stats)) return CC_ERR_INVALID;
    hipLaunchKernelGGL(text_embed_kernel, dim3((e.Bt * e.Lt + 3) / 4), dim3(256), 0, st, e);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_launch_pre_stage(const LnArgs& ln, const TextEmbedArgs& te, float eps, hipStream_t st) {
    if (ln.W > 1024 || (ln.W & 3) || (te.W & 3) || te.W > 1024 || (te.h16 && !te.stats) || (ln.cls && (!ln.pos0 || ln.cls_period <= 0)))
        return CC_ERR_INVALID;
    const int blocks_ln = (ln.rows + 3) / 4;
    hipLaunchKernelGGL(pre_stage_kernel, dim3(blocks_ln + (te.Bt * te.Lt + 3) / 4), dim3(256), 0, st, ln, te, blocks_ln, eps);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_launch_head_project2(const HeadArgs& a0, const HeadArgs* a1, hipStream_t st) {
    if (a0.W > 1024 || (a0.E & 3) || (a1 && (a1->W > 1024 || (a1->E & 3)))) return CC_ERR_UNSUPPORTED;
    if (a0.R + (a1 ? a1->R : 0) > 65535) return CC_ERR_UNSUPPORTED;       // grid y
    HeadPair hp{};
    hp.p[0] = a0;
    hp.p[1] = a1 ? *a1 : a0;
    const int E = (a1 && a1->E > a0.E) ? a1->E : a0.E;
    hipLaunchKernelGGL(head_project_kernel, dim3((E + 63) / 64, a0.R + (a1 ? a1->R : 0)), dim3(256), 0, st, hp, 1e-5f);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_launch_head_project(const float* h, int row_mul, const int* row_idx, const float* gamma, const float* beta,
                           const float* proj, float* out, int R, int W, int E, hipStream_t st) {
    const HeadArgs a{h, row_mul, row_idx, gamma, beta, proj, out, R, W, E};
    return cc_launch_head_project2(a, nullptr, st);
}
