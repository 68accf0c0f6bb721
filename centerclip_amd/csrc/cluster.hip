// Token-cluster hot path for gfx950.
//
//   K0 token_norm_kernel      per-token squared norm / norm / 1/(norm+1e-6)      (HBM stream)
//   K1 gram_dist_kernel       pairwise L2 / cosine distance via exact-f32 MFMA   (MFMA f32)
//      lp_dist_kernel         pairwise Minkowski-p distance, p != 2              (VALU)
//   K2 kmedoids_select_kernel KKZ init + assign/update iterations + sort,        (latency)
//                             one workgroup per problem, D resident in LDS when it fits
//   K3 reduce_tokens_kernel   medoid-token gather / cluster means / pooling + CLS mean (HBM stream)
//
// No host synchronisation anywhere; the [B,K,N,N] temporaries of the reference
// (modules/cluster/fast_kmeans.py:65,81) are never materialised: the update step walks
// per-cluster member lists (SURVEY §8a equivalence 2).
#include "cc_common.h"
#include "cc_kernels.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ============================================================================ K0
// One wave per token.  sqn = sum x^2 (fp32), nrm = sqrtf(sqn), inv = 1/(nrm + 1e-6).
// If xn != nullptr also writes the pre-normalised token x/(nrm+1e-6) to a contiguous
// [P,N,W] buffer (fast_kmeans.py:21-22).
__global__ __launch_bounds__(256) void token_norm_kernel(const float* __restrict__ x, cc_token_layout lay,
                                                         int P, int N, int W, float* __restrict__ sqn,
                                                         float* __restrict__ nrm, float* __restrict__ inv,
                                                         float* __restrict__ xn, int* __restrict__ chunkmax,
                                                         int nchunks) {
    (void)chunkmax; (void)nchunks;                           // (the chunk maxima are per-tile slots now: nothing to reset)
    const int lane = threadIdx.x & 63;
    // problem p on XCD p % 8 (workgroup b runs on XCD b % 8), the mapping of the distance kernels that follow: the
    // tokens this pass pulls into an XCD's L2 are the ones its Gram tiles read next
    const int bpp = (N + 3) >> 2;                                      // workgroups per problem (4 tokens each)
    const int p = ((int)blockIdx.x >> 3) / bpp * 8 + ((int)blockIdx.x & 7);
    const int j = (((int)blockIdx.x >> 3) % bpp) * 4 + (threadIdx.x >> 6);
    if (p >= P || j >= N) return;
    const int tok = p * N + j;
    const float* src = cc_token_ptr(x, lay, p, j);
    float acc = 0.f;
    for (int w = lane * 4; w < W; w += 256) {
        const float4 v = *reinterpret_cast<const float4*>(src + w);
        acc = fmaf(v.x, v.x, acc);
        acc = fmaf(v.y, v.y, acc);
        acc = fmaf(v.z, v.z, acc);
        acc = fmaf(v.w, v.w, acc);
    }
    acc = cc_wave_sum(acc);
    const float n = sqrtf(acc);
    const float r = 1.0f / (n + 1e-6f);
    if (lane == 0) {
        sqn[tok] = acc;
        nrm[tok] = n;
        inv[tok] = r;
    }
    if (xn) {
        float* dst = xn + (int64_t)tok * W;
        const float den = n + 1e-6f;
        for (int w = lane * 4; w < W; w += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + w);
            v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
            *reinterpret_cast<float4*>(dst + w) = v;
        }
    }
}

// (cc_dpp_f32<CTRL>: lane exchange inside a 16-lane DPP row - cc_common.h)

// ============================================================================ K1
// 64x64 tile of the Gram matrix of one problem per 256-thread workgroup (4 waves, 32x32 per wave, 2x2 accumulator
// fragments).  Only tiles with tj >= ti are launched; the transposed tile is written by the same workgroup
// (g_ij == g_ji bit for bit: same k order, commutative products).
//
// Arithmetic: every fp32 token element x is split, while it is staged, into two fp16 values hi = fp16(x) and
// lo = fp16(x - hi) (x = hi + lo to 22 bits), and x.y is accumulated in fp32 as hi.hi + hi.lo + lo.hi on the fp16
// matrix cores (v_mfma_f32_16x16x32_f16: 16x the rate of the exact-fp32 MFMA this kernel used before, which bounded
// it: 2,048 of ~2,400 cycles per k-step).  The dropped lo.lo term is 2^-22 relative, the size of fp32 rounding itself.
// On inputs that fp16 represents exactly (the integer / dyadic lattices of parity levels P1 / P2, the norm-32 tokens
// of the pre_norm fixtures) lo = 0, every product and every partial sum is exact, and D equals any correct fp32
// evaluation bit for bit - the property the index-parity fixtures rest on.  Range: fp16 holds |x| <= 65504 (token
// values of a LayerNorm-ed residual stream are O(1..1e3)); beyond it hi overflows to inf and the distances come out
// inf / NaN.  Below 2^-14 * 2^11 the lo part runs into fp16 subnormals and carries
// fewer bits: the absolute error per element stays <= 2^-25, negligible next to the O(1) elements of the same token.
//
// LDS image per tile: hi and lo planes of [64 rows][64 k] fp16, 128-byte rows, 16-byte chunks XOR-swizzled by (row & 7)
// exactly as the fp16 GEMM stages its operands (conflict-free ds_read_b128 fragment reads).
#define GT 64
#define GK 64
#define CC_METRIC_SQL2 2      /* internal: squared L2 for the spectral affinity (not a clustering metric of the C ABI) */
typedef _Float16 gh8 __attribute__((ext_vector_type(8)));
typedef _Float16 gh4 __attribute__((ext_vector_type(4)));
#define GPLANE (GT * GK)            /* halfs per plane */

// development builds (-DCC_DEV_KNOBS) only: per-workgroup real-time stamps (entry, first stage in LDS, loop done, exit), 100 MHz
#ifdef CC_DEV_KNOBS
__device__ long long* g_gram_prof = nullptr;
#define GRAM_STAMP(slot)                                                                                              \
    do {                                                                                                              \
        if (g_gram_prof && threadIdx.x == 0 && blockIdx.x < 4096) g_gram_prof[(int64_t)blockIdx.x * 4 + (slot)] = (long long)wall_clock64(); \
    } while (0)
#else
#define GRAM_STAMP(slot) do { } while (0)
#endif
template <int METRIC>
__global__ __launch_bounds__(256) void gram_dist_kernel(const float* __restrict__ x, cc_token_layout lay, int N,
                                                        int W, float* sqn, float* nrm, float* inv, int own_norms,
                                                        float* __restrict__ draw, int* __restrict__ chunkmax, int chunk,
                                                        int ntiles, int nprob) {
    // own_norms: the row norms come out of this kernel (sum of squares of the rows it stages anyway; the diagonal
    // tiles publish sqn / nrm / inv for the selection kernel) instead of a separate pass over the tokens (K0).
    extern __shared__ __attribute__((aligned(16))) unsigned char gram_lds_raw[];   // [2 buffers][A,B][hi,lo][GPLANE] halfs + 2 x GT floats
    GRAM_STAMP(0);
    auto plane = [&](int buf, int which, int hl) {
        return reinterpret_cast<_Float16*>(gram_lds_raw) + ((buf * 2 + which) * 2 + hl) * GPLANE;
    };
    // Workgroup b runs on XCD b % 8, each with its own L2: all tiles of a problem are given to ONE XCD (problem
    // p -> XCD p % 8), so a problem's tokens are fetched into one L2 once instead of into up to 8 of them (PMC: the Gram
    // kernel fetched 116 MB for 28.9 MB of tokens with the tile-major order).
    const int tiles_pp = ntiles * (ntiles + 1) / 2;
    const int p = ((int)blockIdx.x >> 3) / tiles_pp * 8 + ((int)blockIdx.x & 7);
    if (p >= nprob) return;
    int t = ((int)blockIdx.x >> 3) % tiles_pp, ti = 0, rowlen = ntiles;
    while (t >= rowlen) { t -= rowlen; ++ti; --rowlen; }
    const int tj = ti + t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lrow = tid >> 4, lchunk = tid & 15;            // 16 threads x 16 B cover one 64-float row slice
    const float* pa[4];
    const float* pb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ra = min(ti * GT + lrow + 16 * q, N - 1);
        const int rb = min(tj * GT + lrow + 16 * q, N - 1);
        pa[q] = cc_token_ptr(x, lay, p, ra) + lchunk * 4;
        pb[q] = cc_token_ptr(x, lay, p, rb) + lchunk * 4;
    }
    const int wr = wave >> 1, wc = wave & 1;
    const bool active = !(ti == tj && wr > wc) && (ti * GT + wr * 32 < N) && (tj * GT + wc * 32 < N);
    const bool diag = (ti == tj);                             // A and B tiles are the same rows: load once
    // 16x16 fragments of this wave's 32x32 block that lie entirely in the padding, or (diagonal block) strictly below
    // the diagonal, are not computed (wave-uniform): the mirrored store of fragment (0,1) covers (1,0)
    const int wrow0 = ti * GT + wr * 32, wcol0 = tj * GT + wc * 32;
    const bool dblock = diag && wr == wc;
    const bool f01 = wcol0 + 16 < N, f10 = (wrow0 + 16 < N) && !dblock, f11 = (wrow0 + 16 < N) && (wcol0 + 16 < N);

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // The tokens come from HBM (first touch) and a k-step's MFMAs are short: with the next slice loaded only one step
    // ahead every step waited most of a memory round trip.  GD register stages: slice kt + GD is requested at the top of
    // step kt into the stage slice kt just left (it went to LDS at the end of step kt - 1).
    constexpr int GD = 3;
    float4 ra_[GD][4], rb_[GD][4];
    const int nk = (W + GK - 1) / GK;
    auto gload = [&](int kt, int st) {
        const bool ok = kt * GK + lchunk * 4 < W;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra_[st][q] = ok ? *reinterpret_cast<const float4*>(pa[q] + kt * GK) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!diag) rb_[st][q] = ok ? *reinterpret_cast<const float4*>(pb[q] + kt * GK) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    float na[4] = {0.f, 0.f, 0.f, 0.f}, nb[4] = {0.f, 0.f, 0.f, 0.f};   // this thread's share of the rows' sums of squares
    auto split_store = [&](_Float16* hi_plane, _Float16* lo_plane, int row, const float4& v) {
        // x = hi + lo, hi = fp16(x), lo = fp16(x - hi); no clamping: beyond fp16's range hi overflows to inf and the
        // distance comes out inf / NaN, as an overflowing fp32 evaluation would (NaN inputs stay NaN)
        const gh4 hi = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        const gh4 lo = {(_Float16)(v.x - (float)hi[0]), (_Float16)(v.y - (float)hi[1]), (_Float16)(v.z - (float)hi[2]),
                        (_Float16)(v.w - (float)hi[3])};
        const int off = row * GK + ((((lchunk >> 1) ^ (row & 7)) << 3) | ((lchunk & 1) << 2));     // halfs
        *reinterpret_cast<gh4*>(hi_plane + off) = hi;
        *reinterpret_cast<gh4*>(lo_plane + off) = lo;
    };
    auto lstore = [&](int buf, int st) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 va = ra_[st][q], vb = rb_[st][q];
            split_store(plane(buf, 0, 0), plane(buf, 0, 1), lrow + 16 * q, va);
            if (!diag) split_store(plane(buf, 1, 0), plane(buf, 1, 1), lrow + 16 * q, vb);
            if (own_norms) {
                na[q] = fmaf(va.x, va.x, na[q]); na[q] = fmaf(va.y, va.y, na[q]);
                na[q] = fmaf(va.z, va.z, na[q]); na[q] = fmaf(va.w, va.w, na[q]);
                if (!diag) {
                    nb[q] = fmaf(vb.x, vb.x, nb[q]); nb[q] = fmaf(vb.y, vb.y, nb[q]);
                    nb[q] = fmaf(vb.z, vb.z, nb[q]); nb[q] = fmaf(vb.w, vb.w, nb[q]);
                }
            }
        }
    };
    gload(0, 0);
#pragma unroll
    for (int u = 1; u < GD; ++u)
        if (u < nk) gload(u, u);
    lstore(0, 0);
    __syncthreads();
    GRAM_STAMP(1);
    const int g = lane >> 4, l15 = lane & 15;
    auto frag = [&](const _Float16* pl, int row, int ks) {     // 8 consecutive k of `row` at k = ks*32 + g*8
        return *reinterpret_cast<const gh8*>(pl + row * GK + ((((ks << 2) | g) ^ (row & 7)) << 3));
    };
    for (int kt0 = 0; kt0 < nk; kt0 += GD) {
#pragma unroll
      for (int u = 0; u < GD; ++u) {
        const int kt = kt0 + u;
        if (kt >= nk) break;
        const int buf = kt & 1;
        if (kt + GD < nk) gload(kt + GD, u);
        if (active) {
            const _Float16* Ah = plane(buf, 0, 0);
            const _Float16* Al = plane(buf, 0, 1);
            const _Float16* Bh = plane(buf, diag ? 0 : 1, 0);
            const _Float16* Bl = plane(buf, diag ? 0 : 1, 1);
            const int ar = wr * 32 + l15, br = wc * 32 + l15;
#pragma unroll
            for (int ks = 0; ks < GK / 32; ++ks) {
                const gh8 a0h = frag(Ah, ar, ks), a0l = frag(Al, ar, ks);
                const gh8 b0h = frag(Bh, br, ks), b0l = frag(Bl, br, ks);
                gh8 a1h = a0h, a1l = a0l, b1h = b0h, b1l = b0l;
                if (f10 || f11) { a1h = frag(Ah, ar + 16, ks); a1l = frag(Al, ar + 16, ks); }
                if (f01 || f11) { b1h = frag(Bh, br + 16, ks); b1l = frag(Bl, br + 16, ks); }
                // acc[fm][fn][r] <-> row i = .. + fm*16 + g*4 + r (A operand rows), column j = .. + fn*16 + l15
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, b0h, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, b0l, acc[0][0], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, b0h, acc[0][0], 0, 0, 0);
                if (f01) {
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, b1h, acc[0][1], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0h, b1l, acc[0][1], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0l, b1h, acc[0][1], 0, 0, 0);
                }
                if (f10) {
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, b0h, acc[1][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, b0l, acc[1][0], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, b0h, acc[1][0], 0, 0, 0);
                }
                if (f11) {
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, b1h, acc[1][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1h, b1l, acc[1][1], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1l, b1h, acc[1][1], 0, 0, 0);
                }
            }
        }
        if (kt + 1 < nk) lstore(buf ^ 1, (u + 1) % GD);
        __syncthreads();
      }
    }

    GRAM_STAMP(2);
    float* lsq = reinterpret_cast<float*>(gram_lds_raw + (size_t)2 * 2 * 2 * GPLANE * sizeof(_Float16));   // [2][GT]
    if (own_norms) {
        // the 16 threads of a row (lchunk) sit in one DPP row: quad swaps, half mirror, mirror
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = na[q], b = nb[q];
            a += cc_dpp_f32<0xB1>(a); b += cc_dpp_f32<0xB1>(b);
            a += cc_dpp_f32<0x4E>(a); b += cc_dpp_f32<0x4E>(b);
            a += cc_dpp_f32<0x141>(a); b += cc_dpp_f32<0x141>(b);
            a += cc_dpp_f32<0x140>(a); b += cc_dpp_f32<0x140>(b);
            if (lchunk == 0) {
                const int r = lrow + 16 * q;
                lsq[r] = a;
                lsq[GT + r] = diag ? a : b;
                const int row = ti * GT + r;
                if (diag && row < N) {                         // every row block has its diagonal tile: publish once
                    const float n = sqrtf(a);
                    sqn[(int64_t)p * N + row] = a;
                    nrm[(int64_t)p * N + row] = n;
                    inv[(int64_t)p * N + row] = 1.0f / (n + 1e-6f);
                }
            }
        }
        __syncthreads();
    }
    // epilogue: distance, chunk max, direct + mirrored store
    float lmax = -3.0e38f;
    if (active) {
        const bool mirror = (tj > ti) || (wc > wr);
        float* Dp = draw + (int64_t)p * N * N;
        const float* sq = sqn + (int64_t)p * N;
        const float* iv = inv + (int64_t)p * N;
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
            for (int fn = 0; fn < 2; ++fn) {
                if ((fm == 0 && fn == 1 && !f01) || (fm == 1 && fn == 0 && !f10) || (fm == 1 && fn == 1 && !f11)) continue;
                const int j = tj * GT + wc * 32 + fn * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = ti * GT + wr * 32 + fm * 16 + g * 4 + r;
                    if (i < N && j < N) {
                        const float gij = acc[fm][fn][r];
                        const float sqi = own_norms ? lsq[i - ti * GT] : sq[i], sqj = own_norms ? lsq[GT + j - tj * GT] : sq[j];
                        float d;
                        if (METRIC == CC_METRIC_EUCLIDEAN) {
                            const float d2 = (sqi + sqj) - 2.0f * gij;
                            d = (i == j) ? 0.0f : sqrtf(fmaxf(d2, 0.0f));
                        } else if (METRIC == CC_METRIC_SQL2) {            // batched_cdist_l2 (cluster_utils.py:121-133):
                            d = (sqj - 2.0f * gij) + sqi;                 // baddbmm(|y|^2, x, y^T, alpha=-2) + |x|^2, unclamped
                        } else {
                            const float ivi = own_norms ? 1.0f / (sqrtf(sqi) + 1e-6f) : iv[i];
                            const float ivj = own_norms ? 1.0f / (sqrtf(sqj) + 1e-6f) : iv[j];
                            d = 1.0f - (gij * ivi) * ivj;
                        }
                        lmax = fmaxf(lmax, d);
                        Dp[(int64_t)i * N + j] = d;
                        if (mirror || (dblock && fm == 0 && fn == 1)) Dp[(int64_t)j * N + i] = d;
                    }
                }
            }
    }
    if (chunkmax) {
        // one slot per (problem, tile, wave), written unconditionally (idle waves: the key below every distance): the
        // selection kernel reduces the slots of its chunk - no reset pass before this kernel, no atomics
        lmax = cc_wave_max(active ? lmax : -3.0e38f);
        if (lane == 0)
            chunkmax[((int64_t)p * tiles_pp + (((int)blockIdx.x >> 3) % tiles_pp)) * 4 + wave] = cc_float_to_ordered_int(lmax);
    }
    GRAM_STAMP(3);
}

// Minkowski-p distance for p != 2 (p == 1 is the shipped MSR-VTT setting, scripts/msrvtt.sh:87).
// Direct sum_w |x-y|^p in ascending w per pair (exact-zero diagonal like ATen's direct path).
// MODE 1: p == 1; MODE 0: general p > 0; MODE 2: p == inf.
#define LLD 36
#define LPK 32    /* k-slice of the VALU kernel */
template <int MODE>
__global__ __launch_bounds__(256) void lp_dist_kernel(const float* __restrict__ x, cc_token_layout lay, int N, int W,
                                                      float pw, float* __restrict__ draw, int* __restrict__ chunkmax,
                                                      int chunk, int ntiles, int nprob) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GT * LLD];      // [buffer][A,B]
    // Workgroup b runs on XCD b % 8, each with its own L2: all tiles of a problem are given to ONE XCD (problem
    // p -> XCD p % 8), so a problem's tokens are fetched into one L2 once instead of into up to 8 of them (PMC: the Gram
    // kernel fetched 116 MB for 28.9 MB of tokens with the tile-major order).
    const int tiles_pp = ntiles * (ntiles + 1) / 2;
    const int p = ((int)blockIdx.x >> 3) / tiles_pp * 8 + ((int)blockIdx.x & 7);
    if (p >= nprob) return;
    int t = ((int)blockIdx.x >> 3) % tiles_pp, ti = 0, rowlen = ntiles;
    while (t >= rowlen) { t -= rowlen; ++ti; --rowlen; }
    const int tj = ti + t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const float* pa[2];
    const float* pb[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pa[q] = cc_token_ptr(x, lay, p, min(ti * GT + lrow + 32 * q, N - 1)) + lchunk * 4;
        pb[q] = cc_token_ptr(x, lay, p, min(tj * GT + lrow + 32 * q, N - 1)) + lchunk * 4;
    }
    const int ty = tid >> 4, tx = tid & 15;      // rows ty + 16*r, cols tx + 16*c
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
    // 16x16 blocks (r, c) of the tile that lie in the padding, or - in a diagonal tile - strictly below the diagonal,
    // are skipped; the mirrored store of block (c, r) covers them
    unsigned live = 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (ti * GT + 16 * r < N && tj * GT + 16 * c < N && !(ti == tj && r > c)) live |= 1u << (4 * r + c);
    const int nk = (W + LPK - 1) / LPK;
    float4 ra_[2], rb_[2];
    auto gload = [&](int kt) {
        const bool ok = kt * LPK + lchunk * 4 < W;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra_[q] = ok ? *reinterpret_cast<const float4*>(pa[q] + kt * LPK) : make_float4(0, 0, 0, 0);
            rb_[q] = ok ? *reinterpret_cast<const float4*>(pb[q] + kt * LPK) : make_float4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<float4*>(&lds[buf][0][(lrow + 32 * q) * LLD + lchunk * 4]) = ra_[q];
            *reinterpret_cast<float4*>(&lds[buf][1][(lrow + 32 * q) * LLD + lchunk * 4]) = rb_[q];
        }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);           // next slice in flight while this one is consumed
#pragma unroll 2
        for (int k4 = 0; k4 < LPK / 4; ++k4) {
            float4 a[4], b[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const float4*>(&lds[buf][0][(ty + 16 * r) * LLD + k4 * 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const float4*>(&lds[buf][1][(tx + 16 * c) * LLD + k4 * 4]);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (!((live >> (4 * r + c)) & 1u)) continue;      // padded or below-diagonal 16x16 block (uniform)
                    const float d0 = fabsf(a[r].x - b[c].x), d1 = fabsf(a[r].y - b[c].y);
                    const float d2 = fabsf(a[r].z - b[c].z), d3 = fabsf(a[r].w - b[c].w);
                    if (MODE == 1) {
                        acc[r][c] = (((acc[r][c] + d0) + d1) + d2) + d3;
                    } else if (MODE == 2) {
                        acc[r][c] = fmaxf(fmaxf(fmaxf(fmaxf(acc[r][c], d0), d1), d2), d3);
                    } else {
                        acc[r][c] = (((acc[r][c] + powf(d0, pw)) + powf(d1, pw)) + powf(d2, pw)) + powf(d3, pw);
                    }
                }
        }
        if (kt + 1 < nk) lstore(buf ^ 1);
        __syncthreads();
    }
    float lmax = -3.0e38f;
    float* Dp = draw + (int64_t)p * N * N;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = ti * GT + ty + 16 * r, j = tj * GT + tx + 16 * c;
            if (((live >> (4 * r + c)) & 1u) && i < N && j < N) {
                float d = acc[r][c];
                if (MODE == 0) d = powf(d, 1.0f / pw);
                lmax = fmaxf(lmax, d);
                Dp[(int64_t)i * N + j] = d;
                if (tj > ti || c > r) Dp[(int64_t)j * N + i] = d;
            }
        }
    if (chunkmax) {
        lmax = cc_wave_max(lmax);
        if (lane == 0)
            chunkmax[((int64_t)p * tiles_pp + (((int)blockIdx.x >> 3) % tiles_pp)) * 4 + (tid >> 6)] = cc_float_to_ordered_int(lmax);
    }
}

// chunkmax[c] = max over the per-(problem, tile, wave) slots of the problems of chunk c (stand-alone distance path; the
// fused path reduces the slots inside the selection kernel).  One workgroup per chunk.
__global__ __launch_bounds__(256) void chunk_max_reduce_kernel(const int* __restrict__ slots, int slots_pp, int P, int chunk,
                                                               int* __restrict__ chunkmax) {
    __shared__ int red[4];
    const int c = blockIdx.x, p0 = c * chunk, p1 = min(P, p0 + chunk);
    int m = (int)0x80000000;
    for (int64_t q = (int64_t)p0 * slots_pp + threadIdx.x; q < (int64_t)p1 * slots_pp; q += 256) m = max(m, slots[q]);
    m = cc_wave_imax(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) chunkmax[c] = max(max(red[0], red[1]), max(red[2], red[3]));
}

// Applies the all_negative shift / self_nearest diagonal to a raw distance tensor
// (cluster_utils.py:35-41) - used by the stand-alone cc_pairwise_distance_f32 only; the
// fused path applies it while staging D into LDS.
__global__ void shift_dist_kernel(float* __restrict__ d, int P, int N, const int* __restrict__ chunkmax, int chunk,
                                  int all_negative, int self_nearest) {
    const int64_t total = (int64_t)P * N * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int p = (int)(idx / ((int64_t)N * N));
        const int rem = (int)(idx - (int64_t)p * N * N);
        const int i = rem / N, j = rem - i * N;
        float v = d[idx];
        if (all_negative) v = (v - cc_ordered_int_to_float(chunkmax[p / chunk])) - 1.0f;
        if (self_nearest && i == j) v -= 1.0f;
        d[idx] = v;
    }
}

// ============================================================================ K2
// One 1024-thread workgroup (16 waves) per problem; the whole selection (KKZ init, assignment / update
// iterations, ascending sort, final assignment) runs inside it with no host round trip.  The staged distance matrix
// takes a whole CU's LDS anyway (154 KB at N = 196), so the workgroup may as well own all of the CU's wave slots:
// staging, the assignment (4 threads per token) and the output phases use all 16 waves; the KKZ chain - K dependent
// arg-max / row-fold steps - runs on ONE wave with no barrier at all (rounds 1-2: four waves meeting in LDS once per step).
// Everything here is latency-bound dependent work on a 38-346 K-entry matrix, so the kernel is
// organised to keep >= 4-8 independent LDS/L2 loads in flight per lane and to avoid LDS
// crossbar shuffles on the critical path (wave arg-max = 4 DPP steps + readlane, ties resolved
// to the lowest index with ballots).  The update step reproduces the association of the reference's
// row sum (ATen's CPU sum, see sum_rank): cluster membership is kept as per-cluster bit masks in
// summation-rank space (ballot over 64 consecutive ranks), from which every iteration derives the
// member list of each cluster in summation order (popcount prefix) together with the tree levels
// each member closes; a candidate then walks its cluster's list with branch-free level partials
// and the medoid falls out of a 64-bit LDS min over (key(sum), token).
//
// LDS carve (dynamic): [IN_LDS: D N*N f32] best K u64 | med K i32 | asg, order N u16
#define SEL_MAX_E 128  /* N <= 8191: 255 passes of 32 terms = up to 16 runs of 16 passes, each folded into the second level of
                        * ATen's row sum when it completes; the third level (pass 256, N = 8,192) is never reached */
#define SEL_MAX_N 8191

// development builds (-DCC_DEV_KNOBS) only: per-problem phase timestamps of K2
#ifdef CC_DEV_KNOBS
__device__ long long* g_sel_prof = nullptr;
#define SEL_PROF_INIT() long long* prof = g_sel_prof
#define SEL_STAMP(slot)                                                      \
    do {                                                                     \
        if (prof && tid == 0) prof[(int64_t)blockIdx.x * 16 + (slot)] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define SEL_PROF_INIT() constexpr long long* prof = nullptr        /* (the stamp code folds away) */
#define SEL_STAMP(slot) do { } while (0)
#endif

// Position of token j in the order in which ATen's CPU row sum (SumKernel.cpp cascade_sum, 8-float vectors; the
// arithmetic of fast_kmeans.py:82) consumes the N terms of a row.  The sum is a fixed tree:
//   total = ((tail + R_0) + R_1 ... + R_7),  R_l = ((P_0l + P_1l) + P_2l) + P_3l,
//   P_kl  = sequential sum over passes it of x[32 it + 8 k + l]  (passes 0..15 and 16.. summed apart, then added),
//   P_0l += the vectors left after the last full pass;  tail = the N % 8 trailing scalars, summed first.
// Masked-out terms are exact zeros, so walking only a cluster's members in this rank order, with a partial sum per
// tree level, gives the reference's value bit for bit.  N < 8 is ATen's scalar path: x0, the scalars from 4 on,
// then x1, x2, x3, strictly left to right.
__device__ __forceinline__ int sum_rank(int j, int N) {
    if (N < 8) {
        const int n4 = N & ~3;
        if (n4 == 0) return j;
        return j == 0 ? 0 : (j >= 4 ? j - 3 : (N - 4) + j);
    }
    const int vec_end = N & ~7, vs = N >> 3, passes = vs >> 2, nleft = vs & 3;
    if (j >= vec_end) return j - vec_end;
    const int l = j & 7, v = j >> 3, k = v & 3, it = v >> 2;
    int off;
    if (v >= 4 * passes) off = passes + (v - 4 * passes);
    else off = (k == 0) ? it : passes + nleft + (k - 1) * passes + it;
    return (N - vec_end) + l * vs + off;
}

// mem[] entry: token (bits 0-12) | flags
#define SEL_TOK 0x1FFFu
#define SEL_F1 0x02000u   /* closes the current run of 16 passes  */
#define SEL_F2 0x04000u   /* closes the current accumulator (k,l) */
#define SEL_F3 0x08000u   /* closes the current lane l            */
#define SEL_TREE 0x10000u /* vector part (not the scalar tail)    */
#define SEL_LEFT 0x20000u /* left-over vector: joins accumulator 0 */
__device__ __forceinline__ float sel_and(float x, int m) { return __int_as_float(__float_as_int(x) & m); }

struct SelSmem {
    float* D;
    unsigned long long* best;   // per cluster: (ordered key of the smallest row sum) << 32 | token, via ds_min_u64
    int* med;
    unsigned short* asg;        // token -> cluster
    unsigned short* order;      // summation rank -> token (see sum_rank)
    unsigned short* prev;       // token -> cluster one iteration ago
    unsigned short* list;       // candidates of the clusters whose member set changed (the only ones whose row sums can differ)
    unsigned char* dirty;       // per cluster: member set changed in this iteration
    int* count;                 // length of list
    // member-list form of the update step (D in global memory only, see the kernel)
    unsigned long long* cmask;  // per cluster: membership bits in summation-rank space
    unsigned int* mem;          // tokens grouped by cluster, each group in summation-rank order (+ the SEL_* flags)
    unsigned short* cnt;        // members per cluster
    unsigned short* off;        // K + 1 group offsets into mem
};

static inline size_t sel_smem_bytes(int N, int K, bool in_lds) {
    size_t b = 0;
    if (in_lds) b += cc_align_up((size_t)N * N * 4, 8);
    b += cc_align_up((size_t)K * 8, 128);       // best (>= 64 B: the chunk-max reduction parks 16 ints at the start)
    b += cc_align_up((size_t)K * 4, 8);         // med
    b += 4 * cc_align_up((size_t)N * 2, 8);     // asg, order, prev, list
    b += cc_align_up((size_t)K, 8) + 8;         // dirty, count
    if (!in_lds) {
        const int E = (N + 63) / 64;
        b += (size_t)K * E * 8;                     // cmask
        b += cc_align_up((size_t)N * 4, 8);         // mem
        b += cc_align_up((size_t)K * 2, 8);         // cnt
        b += cc_align_up((size_t)(K + 1) * 2, 8);   // off
    }
    return cc_align_up(b, 16);
}

__device__ __forceinline__ unsigned cc_wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

__device__ __forceinline__ unsigned cc_wave_umin(unsigned v) {
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xF, 0xF, false));
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

// NE = compile-time number of 64-token chunks (N <= 64*NE): the KKZ loop is fully unrolled, branch free
// What K3 (reduce_tokens_kernel) needs to produce the output rows; kmedoids_select_kernel takes one too and, for the
// shipped variant (medoid gather, W % 32 == 0), writes its problem's 1 + K output rows itself right after the selection -
// the ids never leave LDS and the separate gather launch (10 us of the 99 us op in the step) disappears.
struct GatherDesc {
    const float* x; int64_t in_tok, in_frame;
    int B, T, T_new, n, W, K, mode;
    const long long* medoids; int med_stride;
    const long long* assign;
    const float* cluster_embed;
    const float* cls_mult;
    float* out; int64_t out_tok, out_frame;
    _Float16* h16; float* stats; float* shift;
};
template <bool LEFT>
__device__ void reduce_row_wave(const GatherDesc& g, int row, int lane, const int* med_lds);

// One-iteration-per-launch form (a k-medoids `threshold` above CC_KMEDOIDS_MAX_THRESHOLD: the reference's chunk-mean stop test,
// fast_kmeans.py:85-88, couples the problems of a split chunk, so the chunk is stepped in lockstep from the host side - see
// batch_kmedoids_impl): start from given medoids instead of the KKZ init, pass through untouched once the chunk is done.
struct SelStep {
    const long long* init;       // [P, K] medoids to start from (null: KKZ init)
    const int* done;             // [chunks] != 0: the chunk has passed the stop test - copy init to the output and leave
    int accumulate_iters;        // iters_out[p] += the iterations of this launch
};

#define SEL_WAVES 16
#define SEL_THREADS (64 * SEL_WAVES)
template <bool IN_LDS, int NE>
__global__ __launch_bounds__(SEL_THREADS) void kmedoids_select_kernel(const float* __restrict__ dist_in, float* dist_rw,
                                                              const float* __restrict__ norms,
                                                              const int* __restrict__ chunkmax, int slots_pp, int chunk,
                                                              int apply_shift, int N, int K, int iter_limit,
                                                              int id_sort, long long* __restrict__ medoids_out,
                                                              long long* __restrict__ assign_out,
                                                              int* __restrict__ iters_out, GatherDesc gd, SelStep step) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    if (step.done && step.done[(int)blockIdx.x / chunk]) {         // (workgroup-uniform, before any barrier)
        if (step.init != medoids_out)
            for (int k = threadIdx.x; k < K; k += SEL_THREADS) medoids_out[(int64_t)blockIdx.x * K + k] = step.init[(int64_t)blockIdx.x * K + k];
        return;
    }
    const int E = (N + 63) >> 6;
    SelSmem s;
    {
        unsigned char* q = smem_raw;
        s.D = reinterpret_cast<float*>(q);
        if (IN_LDS) q += cc_align_up((size_t)N * N * 4, 8);
        s.best = reinterpret_cast<unsigned long long*>(q); q += cc_align_up((size_t)K * 8, 128);
        s.med = reinterpret_cast<int*>(q); q += cc_align_up((size_t)K * 4, 8);
        s.asg = reinterpret_cast<unsigned short*>(q); q += cc_align_up((size_t)N * 2, 8);
        s.order = reinterpret_cast<unsigned short*>(q); q += cc_align_up((size_t)N * 2, 8);
        s.prev = reinterpret_cast<unsigned short*>(q); q += cc_align_up((size_t)N * 2, 8);
        s.list = reinterpret_cast<unsigned short*>(q); q += cc_align_up((size_t)N * 2, 8);
        s.dirty = reinterpret_cast<unsigned char*>(q); q += cc_align_up((size_t)K, 8);
        s.count = reinterpret_cast<int*>(q); q += 8;
        s.cmask = reinterpret_cast<unsigned long long*>(q); q += (size_t)K * E * 8;      // (the four below: !IN_LDS only)
        s.mem = reinterpret_cast<unsigned int*>(q); q += cc_align_up((size_t)N * 4, 8);
        s.cnt = reinterpret_cast<unsigned short*>(q); q += cc_align_up((size_t)K * 2, 8);
        s.off = reinterpret_cast<unsigned short*>(q);
    }
    const int p = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t base = (int64_t)p * N * N;
    const float* Dg = dist_in + base;
    SEL_PROF_INIT();
    SEL_STAMP(0);

    // ---- stage D: (d - chunk_max) - 1, then the diagonal - 1 (cluster_utils.py:35-41); 8 loads in flight
    // (when D stays in global memory the shift is applied on the fly by DREAD instead: a read-modify-write pass over
    // the N x N matrix used to be a quarter of this kernel at N = 392-588)
    // torch.max(dis) over the problems of this problem's chunk (cluster_utils.py:36, fast_kmeans.py:127-135): the
    // distance kernel left one maximum per (problem, tile, wave)
    float shift_mx = 0.f;
    if (apply_shift) {
        int* red = reinterpret_cast<int*>(smem_raw);
        const int c0 = (p / chunk) * chunk, c1 = min((int)gridDim.x, c0 + chunk);
        int m = (int)0x80000000;
        for (int64_t q = (int64_t)c0 * slots_pp + tid; q < (int64_t)c1 * slots_pp; q += SEL_THREADS) m = max(m, chunkmax[q]);
        m = cc_wave_imax(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        int mm = red[0];
#pragma unroll
        for (int w = 1; w < SEL_WAVES; ++w) mm = max(mm, red[w]);
        shift_mx = cc_ordered_int_to_float(mm);
        __syncthreads();                                       // red[] is the start of the staged D / of best[]
    }
    if (IN_LDS) {
        const float mx = shift_mx;
        float* dst = IN_LDS ? s.D : (dist_rw + base);
        const int total = N * N;
        if (IN_LDS && (total & 3) == 0) {                      // 16-byte path (problem base stays 16-byte aligned)
            const float4* src4 = reinterpret_cast<const float4*>(Dg);
            float4* dst4 = reinterpret_cast<float4*>(dst);
            const int total4 = total >> 2;
            for (int i0 = tid; i0 < total4; i0 += SEL_THREADS * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * SEL_THREADS;
                    v[u] = idx < total4 ? src4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * SEL_THREADS;
                    if (idx < total4) {
                        if (apply_shift) {
                            v[u].x = (v[u].x - mx) - 1.0f; v[u].y = (v[u].y - mx) - 1.0f;
                            v[u].z = (v[u].z - mx) - 1.0f; v[u].w = (v[u].w - mx) - 1.0f;
                        }
                        dst4[idx] = v[u];
                    }
                }
            }
        } else {
            for (int i0 = tid; i0 < total; i0 += SEL_THREADS * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * SEL_THREADS;
                    v[u] = idx < total ? Dg[idx] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = i0 + u * SEL_THREADS;
                    if (idx < total) dst[idx] = apply_shift ? (v[u] - mx) - 1.0f : v[u];
                }
            }
        }
        if (!IN_LDS) Dg = dist_rw + base;
        __threadfence_block();
        __syncthreads();
        if (apply_shift) {
            float* dd = IN_LDS ? s.D : (dist_rw + base);
            for (int i = tid; i < N; i += SEL_THREADS) dd[(int64_t)i * N + i] -= 1.0f;
            __threadfence_block();
        }
        __syncthreads();
    }
    auto dread_global = [&](int i, int j) {                  // same operations, same order as the staging pass
        float d = Dg[(int64_t)i * N + j];
        if (apply_shift) {
            d = (d - shift_mx) - 1.0f;
            if (i == j) d -= 1.0f;
        }
        return d;
    };
#define DREAD(i, j) (IN_LDS ? s.D[(i) * N + (j)] : dread_global((i), (j)))
    SEL_STAMP(1);

    for (int j = tid; j < N; j += SEL_THREADS) s.order[sum_rank(j, N)] = (unsigned short)j;

    // ---- KKZ init (cluster_utils.py:93,106-118) on ONE wave: lane l owns the TPL consecutive tokens from l * TPL, the
    // running minimum lives in registers as order-preserving uint keys.  Per step: first maximum inside the lane, wave
    // maximum by DPP, lowest lane holding it by one ballot (token order = lane order, so that is the lowest index, as
    // torch.max / argmax resolve ties), its slot by one readlane; then the new medoid's row is folded into the minimum
    // - the lane's TPL entries are contiguous (one ds_read_b128 at N = 196).  No barrier, no LDS exchange in the chain:
    // ~300 cycles per step at N = 196 against 780 for the four-wave form.
    if (step.init) {
        for (int k = tid; k < K; k += SEL_THREADS) s.med[k] = (int)step.init[(int64_t)p * K + k];
    } else if (wave == 0) {
        constexpr int TPL = NE;
        // (the running minimum is kept as plain floats: v_max / v_min / v_cmp_eq order finite floats as torch.max / min do,
        //  -0 == +0 included, and save the 3-instruction key conversion per element that sat in the dependent chain)
        float nearest[TPL];
        const float* nr = norms + (int64_t)p * N;
        const int n0 = lane * TPL;
        const float NEG = -__builtin_inff();                   // padding tokens never win
#pragma unroll
        for (int e = 0; e < TPL; ++e) nearest[e] = (n0 + e < N) ? nr[n0 + e] : NEG;
        for (int i = 0; i < K; ++i) {
            float loc = nearest[0];
            int slot = 0;
#pragma unroll
            for (int e = 1; e < TPL; ++e)
                if (nearest[e] > loc) { loc = nearest[e]; slot = e; }
            float mx = fmaxf(loc, cc_dpp_f32<0xB1>(loc));
            mx = fmaxf(mx, cc_dpp_f32<0x4E>(mx));
            mx = fmaxf(mx, cc_dpp_f32<0x141>(mx));
            mx = fmaxf(mx, cc_dpp_f32<0x140>(mx));
            {
                const int iv = __float_as_int(mx);
                const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
                const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
                mx = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
            }
            const int src = __ffsll((long long)__ballot(loc == mx)) - 1;
            const int m = src * TPL + __builtin_amdgcn_readlane(slot, src);
            if (lane == 0) s.med[i] = m;
            bool done = false;
            if constexpr (IN_LDS && TPL == 4) {
                if ((N & 3) == 0) {                              // 16-byte aligned rows: one wide LDS read
                    const float4 r = *reinterpret_cast<const float4*>(s.D + m * N + min(n0, N - 4));
                    if (n0 < N) {
                        nearest[0] = (i == 0) ? r.x : fminf(nearest[0], r.x);
                        nearest[1] = (i == 0) ? r.y : fminf(nearest[1], r.y);
                        nearest[2] = (i == 0) ? r.z : fminf(nearest[2], r.z);
                        nearest[3] = (i == 0) ? r.w : fminf(nearest[3], r.w);
                    }
                    done = true;
                }
            }
            if (!done) {
                constexpr int RB = TPL <= 16 ? TPL : 8;            // row entries in flight (N > 1,023: TPL up to 64 - the whole
#pragma unroll                                                     //  row at once would not fit the 128 registers of a lane)
                for (int e0 = 0; e0 < TPL; e0 += RB) {
                    float r[RB];
#pragma unroll
                    for (int e = 0; e < RB; ++e) r[e] = (e0 + e < TPL) ? DREAD(m, min(n0 + e0 + e, N - 1)) : 0.f;
#pragma unroll
                    for (int e = 0; e < RB; ++e)
                        if (e0 + e < TPL && n0 + e0 + e < N)
                            nearest[e0 + e] = (i == 0) ? r[e] : fminf(nearest[e0 + e], r[e]);
                }
            }
        }
    }
    __syncthreads();
    SEL_STAMP(2);

    // a_n = first argmin_k D[m_k, n]  (fast_kmeans.py:75-76).  Four threads per token: thread part q scans the contiguous
    // medoid range [q * ceil(K/4), ...) with 8 rows in flight and keeps its first minimum; the quad then keeps the smaller
    // value, the lower cluster id on equal values - the first minimum over all K, as torch.min resolves ties.
    auto assign_step = [&](bool build_masks) {
        if (!IN_LDS && build_masks)                                  // the masks are rebuilt with LDS atomics below
            for (int q = tid; q < K * E; q += SEL_THREADS) s.cmask[q] = 0ull;
        const int part = tid & 3, kq = (K + 3) >> 2;
        const int kb = part * kq, ke = min(K, kb + kq);
        for (int nb = 0; nb < N; nb += SEL_THREADS / 4) {            // uniform trip count
            const int n = nb + (tid >> 2);
            const int nn = min(n, N - 1);
            float best = __builtin_inff();
            int a = K;                                               // (an empty part never wins: D is finite)
            int k = kb;
            for (; k + 8 <= ke; k += 8) {
                int mk[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) mk[u] = s.med[k + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = DREAD(mk[u], nn);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (v[u] < best || a == K) { best = v[u]; a = k + u; }
            }
            for (; k < ke; ++k) {
                const float v = DREAD(s.med[k], nn);
                if (v < best || a == K) { best = v; a = k; }
            }
#pragma unroll
            for (int o = 1; o <= 2; o <<= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oa = __shfl_xor(a, o, 64);
                if (oa < K && (a == K || ob < best || (ob == best && oa < a))) { best = ob; a = oa; }
            }
            if (n < N && part == 0) s.asg[n] = (unsigned short)a;
        }
        if (!IN_LDS && build_masks) {
            // membership bit masks in summation-rank space: bit t of cluster k  <-  token order[t] belongs to k.
            // One ds_or_b64 per token (OR commutes: deterministic) instead of K ballots per wave.
            __syncthreads();
            for (int t = tid; t < N; t += SEL_THREADS)
                atomicOr(&s.cmask[(size_t)s.asg[s.order[t]] * E + (t >> 6)], 1ull << (t & 63));
        }
    };

    int iters = 0;
    long long ph0 = 0, ph1 = 0, ph2 = 0;
    for (int it = 0; it < iter_limit; ++it) {
        long long tq0 = 0, tq1 = 0, tq2 = 0;
        if (prof) tq0 = (long long)__builtin_readcyclecounter();
        assign_step(true);
        __syncthreads();
        if (prof) tq1 = (long long)__builtin_readcyclecounter();
        // Incremental update: s_i depends only on the member set of i's cluster, so a cluster that neither gained nor lost
        // a token since the last iteration keeps its medoid (its best[] entry stays) - only the candidates of the clusters
        // that changed are re-evaluated.  Bit for bit what the full recomputation yields; after the first iteration a
        // handful of clusters move per iteration, and the last iteration (nothing moves) costs an assignment only.
        for (int k = tid; k < K; k += SEL_THREADS) s.dirty[k] = (it == 0) ? 1 : 0;
        if (tid == 0) *s.count = 0;
        __syncthreads();
        for (int n = tid; n < N; n += SEL_THREADS) {
            const unsigned short a = s.asg[n], pa = s.prev[n];
            if (it > 0 && a != pa) { s.dirty[a] = 1; s.dirty[pa] = 1; }
            s.prev[n] = a;
        }
        __syncthreads();
        if constexpr (!IN_LDS) {
        // D in global memory (N > 197): the member-list form - a dense walk reads N^2 entries per iteration from L2,
        // the lists only sum_c |c|^2 (measured at N = 588: 65 k against 37 k cycles per iteration)
        // group the tokens by cluster, each group in summation-rank order: counts -> offsets -> scatter
        for (int k = tid; k < K; k += SEL_THREADS) {
            const unsigned long long* cm = s.cmask + (size_t)k * E;
            int n = 0;
            for (int w = 0; w < E; ++w) n += __popcll(cm[w]);
            s.cnt[k] = (unsigned short)n;
            if (s.dirty[k]) s.best[k] = ~0ull;
        }
        __syncthreads();
        if (wave == 0) {                                       // exclusive prefix sum of the counts: wave scan, 64 at a time
            int carry = 0;
            for (int base = 0; base < K; base += 64) {
                const int v = (base + lane < K) ? (int)s.cnt[base + lane] : 0;
                int incl = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int up = __shfl_up(incl, o, 64);
                    if (lane >= o) incl += up;
                }
                if (base + lane < K) s.off[base + lane] = (unsigned short)(carry + incl - v);
                carry += __shfl(incl, 63, 64);
            }
            if (lane == 0) s.off[K] = (unsigned short)carry;
        }
        __syncthreads();
        const int vec_end = (N < 8) ? 0 : (N & ~7);            // tokens from here on are the sequential tail
        const int rest = (N >> 5) << 5;                        // tokens from here to vec_end: left-over vectors
        // position in ATen's summation tree: lane (bits 7-9) | accumulator (5-6) | run of 16 passes = 512 tokens (0-4;
        // 31 = left-over vector, which joins accumulator 0 behind all of its runs)
        auto tree_code = [&](int j) {
            return ((j & 7) << 7) | (j >= rest ? 31 : ((((j >> 3) & 3) << 5) | (j >> 9)));
        };
        for (int t = tid; t < N; t += SEL_THREADS) {
            const int j = s.order[t];
            const int a = s.asg[j];
            const unsigned long long* cm = s.cmask + (size_t)a * E;
            const int wt = t >> 6;
            unsigned long long below = cm[wt] & ((1ull << (t & 63)) - 1ull);
            int pos = s.off[a] + __popcll(below);
            int tp = below ? 64 * wt + 63 - __clzll((long long)below) : -1;   // rank of the previous member
            for (int w = wt - 1; w >= 0; --w) {
                const unsigned long long m = cm[w];
                pos += __popcll(m);
                if (tp < 0 && m) tp = 64 * w + 63 - __clzll((long long)m);
            }
            // entry = token | which tree levels this member closes (vs the previous member) | tree / left-over flags
            unsigned e = (unsigned)j;
            if (j < vec_end) {
                const int jp = tp >= 0 ? (int)s.order[tp] : -1;
                const int x = (jp < 0 || jp >= vec_end) ? 0x3FF : (tree_code(j) ^ tree_code(jp));
                e |= (x ? SEL_F1 : 0u) | ((x >> 5) ? SEL_F2 : 0u) | ((x >> 7) ? SEL_F3 : 0u) | SEL_TREE |
                     (j >= rest ? SEL_LEFT : 0u);
            }
            s.mem[pos] = e;
        }
        __syncthreads();
        if (prof) tq2 = (long long)__builtin_readcyclecounter();
        // s_i = sum_j D[i,j] * [a_j == a_i]  (fast_kmeans.py:81-82, equivalence 2), evaluated over the members of
        // i's cluster in summation-rank order with one partial sum per level of ATen's tree (see sum_rank):
        // c = current run of passes, P = accumulator (k,l), R = lane l, F = total.  Closing a level adds it to its
        // parent; closing an empty level adds an exact zero, so the level logic is branch-free selects.  The
        // medoid (fast_kmeans.py:82: argmin, lowest index on ties) is a 64-bit LDS min over (key(s_i), i).
        for (int i = tid; i < N; i += SEL_THREADS) {
            const int cl = s.asg[i];
            if (!s.dirty[cl]) continue;
            const int b0 = s.off[cl], b1 = s.off[cl + 1];
            float c = 0.f, P = 0.f, R = 0.f, F = 0.f;
            for (int q = b0; q < b1; q += 8) {
                int e[8];
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) e[u] = (q + u < b1) ? (int)s.mem[q + u] : -1;
#pragma unroll
                for (int u = 0; u < 8; ++u) {                                // padding reads D[i][0], then drops it
                    const float d = DREAD(i, (e[u] < 0 ? 0 : e[u]) & (int)SEL_TOK);
                    v[u] = e[u] < 0 ? 0.f : d;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    // x & m keeps x or yields +0; x - (x & m) is then +0 or x, both exact
                    const int ev = e[u] < 0 ? 0 : e[u];                      // padding: no flags, v = 0
                    const int m1 = -((ev >> 13) & 1), m2 = -((ev >> 14) & 1), m3 = -((ev >> 15) & 1);
                    const int mt = -((ev >> 16) & 1), ml = -((ev >> 17) & 1);
                    const float t1 = sel_and(c, m1);  P += t1;  c -= t1;     // new run of passes / accumulator / lane
                    const float t2 = sel_and(P, m2);  R += t2;  P -= t2;     // new accumulator or lane
                    const float t3 = sel_and(R, m3);  F += t3;  R -= t3;     // new lane
                    const float vt = sel_and(v[u], mt);
                    const float vl = sel_and(vt, ml);
                    P += vl;                                                 // left-over vector -> accumulator 0
                    c += vt - vl;
                    F += v[u] - vt;                                          // tail scalar
                }
            }
            P += c; R += P; F += R;
            atomicMin(&s.best[cl], ((unsigned long long)cc_float_to_ordered_uint(F) << 32) | (unsigned)i);
        }
        __syncthreads();
        } else {
        for (int k = tid; k < K; k += SEL_THREADS)
            if (s.dirty[k]) s.best[k] = ~0ull;
        for (int n = tid; n < N; n += SEL_THREADS)
            if (s.dirty[s.asg[n]]) s.list[atomicAdd(s.count, 1)] = (unsigned short)n;     // (order is irrelevant to the min below)
        __syncthreads();
        if (prof) tq2 = (long long)__builtin_readcyclecounter();
        // s_i = sum_j D[i,j] * [a_j == a_i]  (fast_kmeans.py:81-82, equivalence 2) in the association of ATen's CPU row sum
        // (see sum_rank).  That sum is a fixed TREE, not a chain: 32 accumulators P_kl (lane l = j % 8, accumulator
        // k = (j / 8) % 4), each a sequential sum over the passes it of x[32 it + 8 k + l] (passes 0-15 and 16-31 apart,
        // then added; the up to three vectors behind the last full pass join accumulator 0), then
        // R_l = ((P_0l + P_1l) + P_2l) + P_3l and total = (((tail + R_0) + R_1) ... + R_7), tail = the N % 8 trailing
        // scalars summed first.  The 32 accumulators are independent chains over the passes, walked DENSELY (masked-out
        // terms are exact zeros and x + 0 = x, so adding them is the reference's own arithmetic): depth <= 32 + 3 + 7 + 8
        // dependent adds instead of one per member, no member lists, no per-cluster masks (rounds 1-2: a thread per
        // candidate walking its cluster's member list through branch-free level partials - 9.3 k of the 17.3 k cycles of
        // an iteration at N = 196, set by the largest cluster).  The medoid (fast_kmeans.py:82: argmin,
        // lowest index on ties) is a 64-bit LDS min over (key(s_i), i).
        {
            // 8 lanes per candidate: lane l owns the four accumulators P_0l..P_3l - four independent chains over the passes -
            // forms R_l itself, and the eight R_l meet through one round of lane shuffles; 128 candidates in flight, waves
            // whose eight slots lie behind the end of the list skip the pass.  (A 32-lane form - one accumulator per lane -
            // measured 18.5 k cycles per iteration at N = 196: seven dependent passes of shuffles; a 2-lane form with wide
            // reads 12.5 k: four-way bank conflicts between rows 196 floats apart; this one 9.2 k for all 196 candidates.)
            const int lq = tid & 7, slot = tid >> 3;
            const int M = *s.count;
            if (N >= 8) {
                const int vec_end = N & ~7, vs = N >> 3, passes = vs >> 2, nleft = vs & 3, ntail = N - vec_end;
                for (int i0 = 0; i0 < M; i0 += SEL_THREADS / 8) {      // uniform trip count
                    if (i0 + wave * 8 >= M) continue;                  // wave-uniform
                    const int idx = i0 + slot;
                    const int i = s.list[min(idx, M - 1)];
                    const int cl = s.asg[i];
                    float c0[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
                    const int p0 = min(passes, 16);
#pragma unroll 2
                    for (int it2 = 0; it2 < p0; ++it2) {
                        float d[4];
                        int a4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = 32 * it2 + 8 * k + lq;
                            d[k] = s.D[i * N + j];
                            a4[k] = s.asg[j];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) c0[k] += (a4[k] == cl) ? d[k] : 0.f;
                    }
#pragma unroll 2
                    for (int it2 = 16; it2 < passes; ++it2) {
                        float d[4];
                        int a4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int j = 32 * it2 + 8 * k + lq;
                            d[k] = s.D[i * N + j];
                            a4[k] = s.asg[j];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) c1[k] += (a4[k] == cl) ? d[k] : 0.f;
                    }
                    float P[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) P[k] = (passes > 16) ? c0[k] + c1[k] : c0[k];
                    for (int v = 0; v < nleft; ++v) {                  // the vectors behind the last full pass join accumulator 0
                        const int j = 32 * passes + 8 * v + lq;
                        const float d = s.D[i * N + j];
                        P[0] += ((int)s.asg[j] == cl) ? d : 0.f;
                    }
                    const float R = ((P[0] + P[1]) + P[2]) + P[3];     // R_l
                    float F = 0.f;                                     // tail scalars first (every lane of the group: same value)
                    float dt[7];
                    int at[7];
#pragma unroll
                    for (int t = 0; t < 7; ++t) {
                        const int j = min(vec_end + t, N - 1);
                        dt[t] = s.D[i * N + j];
                        at[t] = s.asg[j];
                    }
#pragma unroll
                    for (int t = 0; t < 7; ++t)
                        if (t < ntail) F += (at[t] == cl) ? dt[t] : 0.f;
                    const int base = (tid & 63) & ~7;
                    float r8[8];
#pragma unroll
                    for (int l = 0; l < 8; ++l) r8[l] = __shfl(R, base + l, 64);
#pragma unroll
                    for (int l = 0; l < 8; ++l) F += r8[l];
                    if (lq == 0 && idx < M)
                        atomicMin(&s.best[cl], ((unsigned long long)cc_float_to_ordered_uint(F) << 32) | (unsigned)i);
                }
            } else {
                // N < 8 is ATen's scalar path: the terms strictly in summation-rank order (x0, the scalars from 4 on, x1..x3)
                for (int q = tid; q < M; q += SEL_THREADS) {
                    const int i = s.list[q];
                    const int cl = s.asg[i];
                    float F = 0.f;
                    for (int t = 0; t < N; ++t) {
                        const int j = s.order[t];
                        const float d = DREAD(i, j);
                        F += (s.asg[j] == cl) ? d : 0.f;
                    }
                    atomicMin(&s.best[cl], ((unsigned long long)cc_float_to_ordered_uint(F) << 32) | (unsigned)i);
                }
            }
        }
        __syncthreads();
        }
        int changed = 0;
        for (int k = tid; k < K; k += SEL_THREADS) {
            const unsigned long long b = s.best[k];
            const int bi = (b == ~0ull) ? 0 : (int)(unsigned)(b & 0xFFFFFFFFull);   // empty cluster -> 0, as argmin
            changed |= (bi != s.med[k]);                                             // over an all-zero row
            s.med[k] = bi;
        }
        ++iters;
        const int any_changed = __syncthreads_or(changed);
        if (prof) { ph0 += tq1 - tq0; ph1 += tq2 - tq1; ph2 += (long long)__builtin_readcyclecounter() - tq2; }
        if (!any_changed) break;
    }
    SEL_STAMP(3);

    if (id_sort) {                              // fast_kmeans.py:90-94 (K <= SEL_THREADS, checked by run_select: one medoid per thread)
        int mine = 0, rank = 0;
        if (tid < K) {
            mine = s.med[tid];
            for (int q = 0; q < K; ++q) {
                const int o = s.med[q];
                rank += (o < mine || (o == mine && q < tid)) ? 1 : 0;
            }
        }
        __syncthreads();
        if (tid < K) s.med[rank] = mine;
        __syncthreads();
        if (assign_out) assign_step(false);
        __syncthreads();
    }
    for (int k = tid; k < K; k += SEL_THREADS) medoids_out[(int64_t)p * K + k] = s.med[k];
    if (assign_out)
        for (int n = tid; n < N; n += SEL_THREADS) assign_out[(int64_t)p * N + n] = (iter_limit > 0 || id_sort) ? s.asg[n] : 0;
    if (iters_out && tid == 0) iters_out[p] = step.accumulate_iters ? iters_out[p] + iters : iters;
    if (gd.out) {
        // K3 folded in: problem p = segment sgm of clip b (p = sgm * B + b, cluster.py:247-250) -> output frame b * T_new + sgm;
        // one wave per output row (CLS mean, K medoid tokens), ids straight from LDS (s.med is final and barrier-visible)
        const int b = p % gd.B, sgm = p / gd.B, Lout = 1 + K;
        const int row0 = (b * gd.T_new + sgm) * Lout;
        for (int l = wave; l < Lout; l += SEL_WAVES) reduce_row_wave<false>(gd, row0 + l, lane, s.med);
    }
    SEL_STAMP(4);
    if (prof && tid == 0) {
        prof[(int64_t)blockIdx.x * 16 + 5] = iters;
        prof[(int64_t)blockIdx.x * 16 + 6] = ph0;
        prof[(int64_t)blockIdx.x * 16 + 7] = ph1;
        prof[(int64_t)blockIdx.x * 16 + 8] = ph2;
    }
#undef DREAD
}

// ============================================================================ K3
// Sum over a strided run of rows in the association of ATen's CPU sum over a non-innermost dimension
// (SumKernel.cpp vectorized_outer_sum -> multi_row_sum): rows are accumulated sequentially, after every 16th row
// the running sum is folded into a second level, after every 256th that one into a third, and the three levels
// are added at the end.  Rows that contribute an exact zero (masked-out tokens) can be skipped.
struct CascadeSum {
    float4 a0, a1, a2;
    int chunk;                                                // 16-row chunk a0 belongs to, -1 = none yet
    __device__ __forceinline__ void init() {
        a0 = a1 = a2 = make_float4(0.f, 0.f, 0.f, 0.f);
        chunk = -1;
    }
    static __device__ __forceinline__ void acc(float4& d, const float4& v) { d.x += v.x; d.y += v.y; d.z += v.z; d.w += v.w; }
    __device__ __forceinline__ void close_chunk(int next_chunk) {
        acc(a1, a0);
        a0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((chunk >> 4) != (next_chunk >> 4)) { acc(a2, a1); a1 = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    __device__ __forceinline__ void add(int row, const float4& v) {
        const int c = row >> 4;
        if (chunk >= 0 && c != chunk) close_chunk(c);
        chunk = c;
        acc(a0, v);
    }
    __device__ __forceinline__ float4 total(int rows) {       // rows = length of the reduced dimension
        if (chunk >= 0 && chunk < (rows >> 4)) {              // the last touched chunk is a complete one
            const int next = ((chunk >> 4) < (rows >> 8)) ? chunk + 16 : chunk;   // its 256-row group complete too?
            close_chunk(next);
        }
        float4 r = a0;
        acc(r, a1);
        acc(r, a2);
        return r;
    }
};

// Sum over the rows for one float4 of columns, in the association ATen uses for THAT column (vectorized_outer_sum):
// columns inside the last full group of 32 -> the plain cascade above; columns behind it (only when W % 32 != 0, which no
// width of the model family has) -> row_sum: four cascades interleaved over the rows (row r -> cascade r % 4 at position
// r / 4), then cascade 0, the up to three left-over rows in order, cascades 1..3.
template <bool LEFT>
struct OuterSum {
    CascadeSum s0;
    CascadeSum s1, s2, s3;                                    // (dead unless LEFT)
    float4 l0, l1, l2;
    bool ilp;
    int rows4;
    __device__ __forceinline__ void init(bool left_columns, int rows) {
        s0.init();
        ilp = LEFT && left_columns;
        rows4 = (rows >> 2) << 2;
        if (LEFT) {
            s1.init(); s2.init(); s3.init();
            l0 = l1 = l2 = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __device__ __forceinline__ void add(int row, const float4& v) {
        if (!LEFT || !ilp) { s0.add(row, v); return; }
        if (row >= rows4) {                                   // left-over rows: kept apart, added one by one in total()
            const int t = row - rows4;
            if (t == 0) l0 = v; else if (t == 1) l1 = v; else l2 = v;
            return;
        }
        const int a = row & 3, r = row >> 2;
        if (a == 0) s0.add(r, v); else if (a == 1) s1.add(r, v); else if (a == 2) s2.add(r, v); else s3.add(r, v);
    }
    __device__ __forceinline__ float4 total(int rows) {
        if (!LEFT || !ilp) return s0.total(rows);
        const int size = rows >> 2;
        float4 r = s0.total(size);
        CascadeSum::acc(r, l0); CascadeSum::acc(r, l1); CascadeSum::acc(r, l2);       // absent rows are exact zeros
        float4 t = s1.total(size); CascadeSum::acc(r, t);
        t = s2.total(size); CascadeSum::acc(r, t);
        t = s3.total(size); CascadeSum::acc(r, t);
        return r;
    }
};

// pairwise_distance(data1, data2) for two DIFFERENT token sets (cluster_utils.py:8-43; the hot path only passes data1 twice):
// one 16x16 output tile per workgroup, both row panels staged through LDS in 64-wide k chunks, plain fp32 VALU.
//   euclidean, p = 2:  sqrt(max(|x|^2 + |y|^2 - 2 x.y, 0))   (ATen's cdist for more than 25 rows; direct below - same value to rounding)
//   euclidean, p != 2: (sum |x - y|^p)^(1/p);   cosine: 1 - (x / (|x| + 1e-6)) . (y / (|y| + 1e-6))
// gmax (optional): the maximum over the whole tensor, as ordered-uint keys (all_negative).
__global__ __launch_bounds__(256) void cross_dist_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int N1,
                                                         int N2, int W, int metric, float p, float* __restrict__ dist,
                                                         unsigned* __restrict__ gmax) {
    __shared__ float a[16][65], b[16][65];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const int bz = blockIdx.z, i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const float* X = x1 + (int64_t)bz * N1 * W;
    const float* Y = x2 + (int64_t)bz * N2 * W;
    const bool gram = metric == CC_METRIC_COSINE || p == 2.0f;
    float dot = 0.f, nx = 0.f, ny = 0.f, lp = 0.f;
    for (int k0 = 0; k0 < W; k0 += 64) {
        for (int e = tid; e < 16 * 64; e += 256) {
            const int r = e >> 6, c = e & 63;
            a[r][c] = (i0 + r < N1 && k0 + c < W) ? X[(int64_t)(i0 + r) * W + k0 + c] : 0.f;
            b[r][c] = (j0 + r < N2 && k0 + c < W) ? Y[(int64_t)(j0 + r) * W + k0 + c] : 0.f;
        }
        __syncthreads();
        const int kn = min(64, W - k0);
        if (gram) {
            for (int c = 0; c < kn; ++c) {
                const float u = a[ti][c], v = b[tj][c];
                dot += u * v; nx += u * u; ny += v * v;
            }
        } else if (p == 1.0f) {
            for (int c = 0; c < kn; ++c) lp += fabsf(a[ti][c] - b[tj][c]);
        } else {
            for (int c = 0; c < kn; ++c) lp += powf(fabsf(a[ti][c] - b[tj][c]), p);
        }
        __syncthreads();
    }
    float d;
    if (metric == CC_METRIC_COSINE) d = 1.0f - dot / ((sqrtf(nx) + 1e-6f) * (sqrtf(ny) + 1e-6f));
    else if (p == 2.0f) d = sqrtf(fmaxf((nx + ny) - 2.f * dot, 0.f));
    else d = (p == 1.0f) ? lp : powf(lp, 1.0f / p);
    const bool in = i0 + ti < N1 && j0 + tj < N2;
    if (in) dist[((int64_t)bz * N1 + i0 + ti) * N2 + j0 + tj] = d;
    if (gmax) {
        float m = in ? d : -__builtin_inff();
        m = cc_wave_max(m);
        if ((tid & 63) == 0) atomicMax(gmax, cc_float_to_ordered_uint(m));
    }
}

// dis = dis - max(dis) - 1 (all_negative, :35-36), then dis[..., j, j] -= 1 for j < N2 (self_nearest, :38-41)
__global__ __launch_bounds__(256) void cross_shift_kernel(float* __restrict__ dist, int P, int N1, int N2,
                                                          const unsigned* __restrict__ gmax, int self_nearest) {
    float mx = 0.f;
    if (gmax) {
        const unsigned u = *gmax;
        mx = __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
    }
    const int64_t total = (int64_t)P * N1 * N2;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int j = (int)(idx % N2);
        const int i = (int)((idx / N2) % N1);
        float v = dist[idx];
        if (gmax) v = (v - mx) - 1.0f;
        if (self_nearest && i == j) v -= 1.0f;
        dist[idx] = v;
    }
}

// One wave per output row.  mode 0: K medoid tokens (cluster.py:289; med_stride 0 = the same ids for every problem,
// 'sparse_sampling' :326-343); mode 1: K cluster means (cluster.py:291-301:
// sum(res_tmp * mask, dim=1) / sum(mask), empty cluster -> 0/0 = NaN as in the reference); mode 2: 'pooling'
// (cluster.py:319-324: every token, CLS included, = mean over the segment's frames).  Row 0 of modes 0/1 is the mean
// of the segment's CLS tokens (cluster.py:307-308), optionally scaled per frame (adaptive_cls, :244-245);
// cluster_embed [K,W] is added to rows 1..K (:304-305).
template <bool LEFT>
__device__ void reduce_row_wave(const GatherDesc& gd, int row, int lane, const int* med_lds) {
    const float* __restrict__ x = gd.x;
    const int64_t in_tok = gd.in_tok, in_frame = gd.in_frame, out_tok = gd.out_tok, out_frame = gd.out_frame;
    const int B = gd.B, T = gd.T, T_new = gd.T_new, n = gd.n, W = gd.W, K = gd.K, mode = gd.mode;
    const long long* __restrict__ medoids = gd.medoids;
    const int med_stride = gd.med_stride;
    const long long* __restrict__ assign = gd.assign;
    const float* __restrict__ cluster_embed = gd.cluster_embed;
    const float* __restrict__ cls_mult = gd.cls_mult;
    float* __restrict__ out = gd.out;
    _Float16* __restrict__ h16 = gd.h16;
    float* __restrict__ stats = gd.stats;
    float* __restrict__ shift = gd.shift;
    const int Lout = (mode == 2) ? 1 + n : 1 + K;
    // optional by-products for the fused forward (the next block's folded ln_1): fp16 copy of the output row at
    // h16[row][W], centred on the row mean (see layernorm_row in transformer.hip), its (sum, sum of squares) and the mean.
    // The mean needs the whole row: every lane re-reads the elements it has just written (same lane, program order).
    float st_t = 0.f;
    float* own_row = nullptr;
    auto emit = [&](int w, const float4& v) {
        (void)w;
        if (h16) st_t += (v.x + v.y) + (v.z + v.w);
    };
    auto finish = [&]() {
        if (!h16) return;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const float om = cc_wave_sum(st_t) / (float)W;
        float st_s = 0.f, st_q = 0.f;
        for (int w = lane * 4; w < W; w += 256) {
            const float4 v = *reinterpret_cast<const float4*>(own_row + w);
            h4 o = {(_Float16)(v.x - om), (_Float16)(v.y - om), (_Float16)(v.z - om), (_Float16)(v.w - om)};
            *reinterpret_cast<h4*>(h16 + (int64_t)row * W + w) = o;
            const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
            st_s += (q0 + q1) + (q2 + q3);
            st_q += (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
        }
        st_s = cc_wave_sum(st_s);
        st_q = cc_wave_sum(st_q);
        if (lane == 0) {
            reinterpret_cast<float2*>(stats)[row] = make_float2(st_s, st_q);
            if (shift) shift[row] = om;
        }
    };
    const int seg = row / Lout, l = row - seg * Lout;
    const int b = seg / T_new, sgm = seg - b * T_new;
    const int fd = T / T_new, N = fd * n;
    float* dst = out + (int64_t)l * out_tok + (int64_t)seg * out_frame;
    own_row = dst;
    const float* seg0 = x + (int64_t)(b * T + sgm * fd) * in_frame;       // CLS token of the segment's first frame
    if (l == 0 || mode == 2) {
        const float* src = seg0 + (int64_t)l * in_tok;
        const float den = (float)fd;
        for (int w = lane * 4; w < W; w += 256) {
            OuterSum<LEFT> cs;
            cs.init(w >= (W & ~31), fd);
            for (int f = 0; f < fd; ++f) {
                float4 v = *reinterpret_cast<const float4*>(src + (int64_t)f * in_frame + w);
                if (cls_mult && l == 0 && mode != 2) {
                    const float m = cls_mult[sgm * fd + f];
                    v.x *= m; v.y *= m; v.z *= m; v.w *= m;
                }
                cs.add(f, v);
            }
            float4 a = cs.total(fd);
            a.x = a.x / den; a.y = a.y / den; a.z = a.z / den; a.w = a.w / den;
            *reinterpret_cast<float4*>(dst + w) = a;
            emit(w, a);
        }
        finish();
        return;
    }
    const int p = sgm * B + b, k = l - 1;
    const float* emb = cluster_embed ? cluster_embed + (int64_t)k * W : nullptr;
    if (mode == 0) {
        const int j = med_lds ? med_lds[k] : (int)medoids[(int64_t)p * med_stride + k];
        const int f = j / n, i = j - f * n;
        const float* src = seg0 + (int64_t)(1 + i) * in_tok + (int64_t)f * in_frame;
        for (int w = lane * 4; w < W; w += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + w);
            if (emb) {
                const float4 e = *reinterpret_cast<const float4*>(emb + w);
                v.x += e.x; v.y += e.y; v.z += e.z; v.w += e.w;
            }
            *reinterpret_cast<float4*>(dst + w) = v;
            emit(w, v);
        }
        finish();
        return;
    }
    // mode 1: members of cluster k in ascending token order (64 tokens per ballot)
    const long long* as = assign + (int64_t)p * N;
    int count = 0;
    for (int j0 = 0; j0 < N; j0 += 64) count += __popcll(__ballot(j0 + lane < N && as[j0 + lane] == k));
    const float den = (float)count;
    for (int w0 = 0; w0 < W; w0 += 256) {                         // wave-uniform trip count; lanes past W idle
        const int w = w0 + lane * 4;
        OuterSum<LEFT> cs;
        cs.init(w >= (W & ~31), N);
        for (int j0 = 0; j0 < N; j0 += 64) {
            unsigned long long m = __ballot(j0 + lane < N && as[j0 + lane] == k);
            while (m) {
                const int j = j0 + __ffsll((long long)m) - 1;
                m &= m - 1ull;
                const int f = j / n, i = j - f * n;
                if (w < W)
                    cs.add(j, *reinterpret_cast<const float4*>(seg0 + (int64_t)(1 + i) * in_tok + (int64_t)f * in_frame + w));
            }
        }
        if (w < W) {
            float4 a = cs.total(N);
            a.x = a.x / den; a.y = a.y / den; a.z = a.z / den; a.w = a.w / den;
            if (emb) {
                const float4 e = *reinterpret_cast<const float4*>(emb + w);
                a.x += e.x; a.y += e.y; a.z += e.z; a.w += e.w;
            }
            *reinterpret_cast<float4*>(dst + w) = a;
            emit(w, a);
        }
    }
    finish();
}

template <bool LEFT>
__global__ __launch_bounds__(256) void reduce_tokens_kernel(GatherDesc gd) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Lout = (gd.mode == 2) ? 1 + gd.n : 1 + gd.K;
    if (row >= gd.B * gd.T_new * Lout) return;
    reduce_row_wave<LEFT>(gd, row, threadIdx.x & 63, nullptr);
}

// ============================================================================ host side
static GatherDesc gather_desc(const float* x, int64_t in_tok, int64_t in_frame, int B, int T, int T_new, int n, int W, int K,
                              int mode, const long long* medoids, int med_stride, const long long* assign,
                              const float* cluster_embed, const float* cls_mult, float* out, int64_t out_tok,
                              int64_t out_frame, _Float16* h16, float* stats, float* shift) {
    GatherDesc g{};
    g.x = x; g.in_tok = in_tok; g.in_frame = in_frame;
    g.B = B; g.T = T; g.T_new = T_new; g.n = n; g.W = W; g.K = K; g.mode = mode;
    g.medoids = medoids; g.med_stride = med_stride; g.assign = assign;
    g.cluster_embed = cluster_embed; g.cls_mult = cls_mult;
    g.out = out; g.out_tok = out_tok; g.out_frame = out_frame;
    g.h16 = h16; g.stats = stats; g.shift = shift;
    return g;
}

// K3 as a launch of its own (one wave per output row)
static void launch_reduce_tokens(hipStream_t st, const float* x, int64_t in_tok, int64_t in_frame, int B, int T, int T_new,
                                 int n, int W, int K, int mode, const long long* medoids, int med_stride,
                                 const long long* assign, const float* cluster_embed, const float* cls_mult, float* out,
                                 int64_t out_tok, int64_t out_frame, _Float16* h16, float* stats, float* shift) {
    const GatherDesc g = gather_desc(x, in_tok, in_frame, B, T, T_new, n, W, K, mode, medoids, med_stride, assign,
                                     cluster_embed, cls_mult, out, out_tok, out_frame, h16, stats, shift);
    const int rows = B * T_new * ((mode == 2) ? 1 + n : 1 + K);
    if (W & 31) hipLaunchKernelGGL(reduce_tokens_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, st, g);
    else hipLaunchKernelGGL(reduce_tokens_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, st, g);
}

namespace {

struct ClusterWs {
    int* tilemax;
    int slots_pp;
    int* chunkmax;
    float* sqn;
    float* nrm;
    float* inv;
    float* draw;
    float* xn;
    long long* med;
    long long* asg;
    long long* med2;     // second medoid buffer + per-chunk stop flags of the stepped (loose threshold) selection
    int* done;
    size_t total;
};

// extra scratch of the spectral selection: L_sym [P,N,N], the eigensolver's global copy (N > 201), Q [P,N,K4], and the
// k-medoids scratch for P problems of N K4-wide rows (pre-normalised)
struct SpectralWs {
    float* lap;
    float* q;
    int k4;
    void* eig;
    size_t eig_bytes;
    void* km;
    size_t km_bytes;
    size_t total;
};
ClusterWs carve(void* ws, int P, int N, int W, int pre_norm, int K_for_med);
extern "C" size_t cc_spectral_embedding_workspace_bytes(int32_t P, int32_t N);
SpectralWs carve_spectral(void* ws, int P, int N, int K) {
    SpectralWs s{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* ptr = ws ? static_cast<char*>(ws) + off : nullptr;
        off += cc_align_up(bytes, 256);
        return ptr;
    };
    s.k4 = (K + 3) / 4 * 4;
    s.lap = static_cast<float*>(take((size_t)P * N * N * sizeof(float)));
    s.q = static_cast<float*>(take((size_t)P * N * s.k4 * sizeof(float)));
    s.eig_bytes = cc_spectral_embedding_workspace_bytes(P, N);
    s.eig = take(s.eig_bytes);
    s.km_bytes = carve(nullptr, P, N, s.k4, 1, N).total;
    s.km = take(s.km_bytes);
    s.total = off;
    return s;
}

ClusterWs carve(void* ws, int P, int N, int W, int pre_norm, int K_for_med) {
    ClusterWs c{};
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void* ptr = ws ? static_cast<char*>(ws) + off : nullptr;
        off += cc_align_up(bytes, 256);
        return ptr;
    };
    {   // per-(problem, tile, wave) maxima of the distance kernel + the reduced per-chunk values (stand-alone path)
        const size_t nt = (size_t)(N + GT - 1) / GT;
        c.slots_pp = (int)(nt * (nt + 1) / 2 * 4);
        c.tilemax = static_cast<int*>(take((size_t)P * c.slots_pp * 4));
        c.chunkmax = static_cast<int*>(take((size_t)P * 4));
    }
    c.sqn = static_cast<float*>(take((size_t)P * N * 4));
    c.nrm = static_cast<float*>(take((size_t)P * N * 4));
    c.inv = static_cast<float*>(take((size_t)P * N * 4));
    c.draw = static_cast<float*>(take((size_t)P * N * N * 4));
    c.med = static_cast<long long*>(take((size_t)P * (size_t)K_for_med * 8));
    c.asg = static_cast<long long*>(take((size_t)P * N * 8));
    c.med2 = static_cast<long long*>(take((size_t)P * (size_t)K_for_med * 8));
    c.done = static_cast<int*>(take((size_t)P * 4));
    c.xn = pre_norm ? static_cast<float*>(take((size_t)P * N * W * 4)) : nullptr;
    c.total = off;
    return c;
}

bool layout_ok(const cc_token_layout* l, int W) {
    if (!l || l->B <= 0 || l->S <= 0 || l->fd <= 0 || l->n <= 0) return false;
    if (W <= 0 || (W & 3)) return false;
    return !((l->stride_b | l->stride_s | l->stride_f | l->stride_i) & 3);
}

cc_token_layout contiguous_layout(int P, int N, int W) {
    cc_token_layout l;
    l.B = P; l.S = 1; l.fd = 1; l.n = N;
    l.stride_b = (int64_t)N * W; l.stride_s = 0; l.stride_f = 0; l.stride_i = W;
    return l;
}

// K0 (+ optional pre-norm copy) and K1: raw distances + chunk max in ws
int run_distance(const float* x, cc_token_layout lay, int W, int metric, float p, int chunk, int pre_norm,
                 const ClusterWs& c, hipStream_t st) {
    const int P = lay.B * lay.S, N = lay.fd * lay.n;
    const int nb = ((P + 7) / 8) * 8 * ((N + 3) / 4);                  // token_norm_kernel: problem p on XCD p % 8
    if (pre_norm) {
        hipLaunchKernelGGL(token_norm_kernel, dim3(nb), dim3(256), 0, st, x, lay, P, N, W, c.sqn, c.nrm, c.inv, c.xn,
                           (int*)nullptr, 0);
        x = c.xn;
        lay = contiguous_layout(P, N, W);
    }
    // the Gram kernels produce the row norms themselves (of the pre-normalised copy when pre_norm made one above);
    // the Minkowski kernels (no Gram) keep the separate norm pass
    const bool own_norms = (metric == CC_METRIC_COSINE || p == 2.0f);
    if (!own_norms) {
        hipLaunchKernelGGL(token_norm_kernel, dim3(nb), dim3(256), 0, st, x, lay, P, N, W, c.sqn, c.nrm, c.inv,
                           (float*)nullptr, (int*)nullptr, 0);
        CC_LAUNCH_CHECK();
    }
    const int nt = (N + GT - 1) / GT;
    dim3 grid((unsigned)(((P + 7) / 8) * 8 * (nt * (nt + 1) / 2)));       // 1-D: problem p on XCD p % 8
    const size_t gram_smem = (size_t)2 * 2 * 2 * GT * GK * sizeof(_Float16) + 2 * GT * sizeof(float);   // 66,048 B
    if (metric == CC_METRIC_COSINE || p == 2.0f) {
        static bool configured = false;
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gram_dist_kernel<CC_METRIC_COSINE>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_smem) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(gram_dist_kernel<CC_METRIC_EUCLIDEAN>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_smem) != hipSuccess)
                return CC_ERR_HIP;
            configured = true;
        }
    }
    if (metric == CC_METRIC_COSINE) {
        hipLaunchKernelGGL(gram_dist_kernel<CC_METRIC_COSINE>, grid, dim3(256), gram_smem, st, x, lay, N, W, c.sqn, c.nrm,
                           c.inv, own_norms ? 1 : 0, c.draw, c.tilemax, chunk, nt, P);
    } else if (p == 2.0f) {
        hipLaunchKernelGGL(gram_dist_kernel<CC_METRIC_EUCLIDEAN>, grid, dim3(256), gram_smem, st, x, lay, N, W, c.sqn,
                           c.nrm, c.inv, own_norms ? 1 : 0, c.draw, c.tilemax, chunk, nt, P);
    } else if (p == 1.0f) {
        hipLaunchKernelGGL(lp_dist_kernel<1>, grid, dim3(256), 0, st, x, lay, N, W, p, c.draw, c.tilemax, chunk, nt, P);
    } else if (p > 3.0e38f) {
        hipLaunchKernelGGL(lp_dist_kernel<2>, grid, dim3(256), 0, st, x, lay, N, W, p, c.draw, c.tilemax, chunk, nt, P);
    } else {
        hipLaunchKernelGGL(lp_dist_kernel<0>, grid, dim3(256), 0, st, x, lay, N, W, p, c.draw, c.tilemax, chunk, nt, P);
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// raw squared-L2 distances (spectral affinity): the Gram kernel with its own row norms, no shift
int run_distance_sq(const float* x, cc_token_layout lay, int W, const ClusterWs& c, hipStream_t st) {
    const int P = lay.B * lay.S, N = lay.fd * lay.n;
    const int nt = (N + GT - 1) / GT;
    dim3 grid((unsigned)(((P + 7) / 8) * 8 * (nt * (nt + 1) / 2)));
    const size_t gram_smem = (size_t)2 * 2 * 2 * GT * GK * sizeof(_Float16) + 2 * GT * sizeof(float);
    static bool configured = false;
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gram_dist_kernel<CC_METRIC_SQL2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)gram_smem) != hipSuccess)
            return CC_ERR_HIP;
        configured = true;
    }
    hipLaunchKernelGGL(gram_dist_kernel<CC_METRIC_SQL2>, grid, dim3(256), gram_smem, st, x, lay, N, W, c.sqn, c.nrm, c.inv,
                       1, c.draw, c.tilemax, P, nt, P);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int run_select(const float* dist_in, float* dist_rw, const float* norms, const int* chunkmax, int slots_pp, int chunk,
               int apply_shift, int P, int N, int K, int iter_limit, int id_sort, long long* med, long long* assign,
               int* iters, hipStream_t st, const GatherDesc* gather = nullptr, const SelStep* step = nullptr) {
    GatherDesc gd{};
    if (gather) gd = *gather;
    SelStep ss{};
    if (step) ss = *step;
    const size_t lds_limit = 160 * 1024;
    const bool in_lds = sel_smem_bytes(N, K, true) <= lds_limit;
    const size_t smem = sel_smem_bytes(N, K, in_lds);
    if (smem > lds_limit || K > SEL_THREADS) return CC_ERR_UNSUPPORTED;   // (per-cluster masks K x ceil(N / 64) x 8 bytes in LDS)
    const int ne = (N + 63) / 64;
#define SEL_LAUNCH(INLDS, NEV)                                                                                         \
    do {                                                                                                               \
        auto kern = kmedoids_select_kernel<INLDS, NEV>;                                                                \
        if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return CC_ERR_HIP;                                                                                         \
        hipLaunchKernelGGL(kern, dim3(P), dim3(SEL_THREADS), smem, st, dist_in, dist_rw, norms, chunkmax, slots_pp, chunk, apply_shift, \
                           N, K, iter_limit, id_sort, med, assign, iters, gd, ss);                                    \
    } while (0)
    if (in_lds) {
        if (ne <= 1) SEL_LAUNCH(true, 1);
        else if (ne == 2) SEL_LAUNCH(true, 2);
        else if (ne == 3) SEL_LAUNCH(true, 3);
        else SEL_LAUNCH(true, 4);
    } else {
        if (ne <= 4) SEL_LAUNCH(false, 4);
        else if (ne <= 7) SEL_LAUNCH(false, 7);
        else if (ne <= 10) SEL_LAUNCH(false, 10);
        else if (ne <= 16) SEL_LAUNCH(false, 16);
        else if (ne <= 25) SEL_LAUNCH(false, 25);           // N <= 1,600: ViT-B/16, 8 frames per segment (1,568)
        else if (ne <= 40) SEL_LAUNCH(false, 40);
        else if (ne <= 64) SEL_LAUNCH(false, 64);
        else SEL_LAUNCH(false, SEL_MAX_E);
    }
#undef SEL_LAUNCH
    CC_LAUNCH_CHECK();
    return CC_OK;
}

// ---- the reference's stop test, literally (fast_kmeans.py:85-88), for thresholds the fixed-point test cannot stand in for.
// ATen's CPU sum of n fp32 terms along a contiguous dimension (SumKernel.cpp cascade_sum; the same tree as sum_rank above,
// restated for a dense walk): 8 lanes x 4 interleaved accumulators over passes of 32 terms, level 0 folded into
// level 1 every 16 passes, left-over vectors into accumulator 0, accumulators 1..3 added to 0, the scalar tail first, then the
// lanes 0..7.  One thread walks the whole tree (n < 8,192: the third level is never reached).
template <typename F>
__device__ float aten_row_sum(int n, F term) {
    if (n < 8) {                                              // the same scheme on scalars
        float q[4] = {0.f, 0.f, 0.f, 0.f};
        const int s4 = n >> 2;
        for (int i = 0; i < s4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] += term(4 * i + k);
        for (int t = 4 * s4; t < n; ++t) q[0] += term(t);
        q[0] += q[1]; q[0] += q[2]; q[0] += q[3];
        return q[0];
    }
    const int vs = n >> 3, passes = vs >> 2;
    float a0[32], a1[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) { a0[u] = 0.f; a1[u] = 0.f; }
    int i = 0;
    while (i + 16 <= passes) {
        for (int r = 0; r < 16; ++r, ++i)
#pragma unroll
            for (int u = 0; u < 32; ++u) a0[u] += term(32 * i + u);
#pragma unroll
        for (int u = 0; u < 32; ++u) { a1[u] += a0[u]; a0[u] = 0.f; }
    }
    for (; i < passes; ++i)
#pragma unroll
        for (int u = 0; u < 32; ++u) a0[u] += term(32 * i + u);
#pragma unroll
    for (int u = 0; u < 32; ++u) a0[u] += a1[u];
    for (int v = 4 * passes; v < vs; ++v)
#pragma unroll
        for (int l = 0; l < 8; ++l) a0[l] += term(8 * v + l);
    float out = 0.f;
    for (int t = 8 * vs; t < n; ++t) out += term(t);
#pragma unroll
    for (int l = 0; l < 8; ++l) out += ((a0[l] + a0[8 + l]) + a0[16 + l]) + a0[24 + l];
    return out;
}

// center_shift of one split chunk per workgroup: mean over the chunk's problems of sum_k |X[m_k] - X[m_k_prev]|_2, each of the
// three sums in ATen's association; done[chunk] = 1 when it falls below the threshold (the reference's `break`).
__global__ __launch_bounds__(256) void center_shift_kernel(const float* __restrict__ x, cc_token_layout lay, int W, int P, int K,
                                                           int chunk, const long long* __restrict__ med_new,
                                                           const long long* __restrict__ med_prev, float threshold,
                                                           int* __restrict__ done, float* __restrict__ rows_ws) {
    const int c = blockIdx.x, p0 = c * chunk, bc = min(chunk, P - p0);
    if (done[c]) return;
    float* rows = rows_ws + (int64_t)p0 * K;                  // [bc, K] scratch (global: K * chunk is unbounded)
    for (int r = threadIdx.x; r < bc * K; r += 256) {
        const int b = r / K, k = r - b * K, p = p0 + b;
        const float* xa = cc_token_ptr(x, lay, p, (int)med_new[(int64_t)p * K + k]);
        const float* xb = cc_token_ptr(x, lay, p, (int)med_prev[(int64_t)p * K + k]);
        const float ss = aten_row_sum(W, [&](int j) { const float d = xa[j] - xb[j]; return d * d; });
        rows[r] = sqrtf(ss);
    }
    __threadfence_block();
    __syncthreads();
    __shared__ float per_problem[1024];                       // (bc <= 1,024: checked by the caller)
    for (int b = threadIdx.x; b < bc; b += 256)
        per_problem[b] = aten_row_sum(K, [&](int k) { return rows[(int64_t)b * K + k]; });
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = aten_row_sum(bc, [&](int b) { return per_problem[b]; });
        const float shift = total / (float)bc;
        if (shift < threshold) done[c] = 1;
    }
}

bool p_supported(int metric, float p) { return metric == CC_METRIC_COSINE || (p > 0.0f); }

}  // namespace

extern "C" {

#ifdef CC_DEV_KNOBS
int cc_debug_set_gram_profile(long long* buf) {     // development builds only; buf [4096, 4] int64 device memory or NULL
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gram_prof), &buf, sizeof(buf)) == hipSuccess ? CC_OK : CC_ERR_HIP;
}
int cc_debug_set_select_profile(long long* buf) {   // development builds only; buf [P,16] int64 device memory or NULL
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sel_prof), &buf, sizeof(buf)) == hipSuccess ? CC_OK : CC_ERR_HIP;
}
#endif

size_t cc_spectral_workspace_bytes(int32_t P, int32_t N, int32_t K) {
    if (P <= 0 || N <= 0 || K <= 0) return 0;
    return carve_spectral(nullptr, P, N, K).total;
}

size_t cc_cluster_workspace_bytes(int32_t P, int32_t N, int32_t W, int32_t pre_norm) {
    if (P <= 0 || N <= 0 || W <= 0) return 0;
    return carve(nullptr, P, N, W, pre_norm, N).total;
}

int cc_token_norms_f32(const float* x, const cc_token_layout* lay, int32_t W, float* norms, void* ws,
                       size_t ws_bytes, void* stream) {
    if (!x || !norms || !layout_ok(lay, W)) return CC_ERR_INVALID;
    const int P = lay->B * lay->S, N = lay->fd * lay->n;
    ClusterWs c = carve(ws, P, N, W, 0, N);
    if (!ws || ws_bytes < c.total) return CC_ERR_WORKSPACE;
    hipLaunchKernelGGL(token_norm_kernel, dim3(((P + 7) / 8) * 8 * ((N + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, *lay,
                       P, N, W, c.sqn, norms, c.inv, (float*)nullptr, (int*)nullptr, 0);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_pairwise_distance_f32(const float* x, const cc_token_layout* lay, int32_t W, int32_t metric, float p,
                             int32_t all_negative, int32_t self_nearest, int32_t chunk, float* dist,
                             float* norms_out, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !dist || !layout_ok(lay, W)) return CC_ERR_INVALID;
    if (metric != CC_METRIC_EUCLIDEAN && metric != CC_METRIC_COSINE) return CC_ERR_UNSUPPORTED;
    if (!p_supported(metric, p)) return CC_ERR_UNSUPPORTED;
    const int P = lay->B * lay->S, N = lay->fd * lay->n;
    if (chunk <= 0) chunk = P;
    ClusterWs c = carve(ws, P, N, W, 0, N);
    if (!ws || ws_bytes < c.total) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    c.draw = dist;
    if (norms_out) c.nrm = norms_out;
    int rc = run_distance(x, *lay, W, metric, p, chunk, 0, c, st);
    if (rc != CC_OK) return rc;
    if (all_negative || self_nearest) {
        const int64_t total = (int64_t)P * N * N;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        if (all_negative) {
            hipLaunchKernelGGL(chunk_max_reduce_kernel, dim3((P + chunk - 1) / chunk), dim3(256), 0, st, c.tilemax, c.slots_pp,
                               P, chunk, c.chunkmax);
            CC_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(shift_dist_kernel, dim3(blocks), dim3(256), 0, st, dist, P, N, c.chunkmax, chunk,
                           all_negative, self_nearest);
        CC_LAUNCH_CHECK();
    }
    return CC_OK;
}

int cc_pairwise_distance_cross_f32(const float* x1, const float* x2, int32_t P, int32_t N1, int32_t N2, int32_t W,
                                   int32_t metric, float p, int32_t all_negative, int32_t self_nearest, float* dist,
                                   void* ws, size_t ws_bytes, void* stream) {
    if (!x1 || !x2 || !dist || P <= 0 || N1 <= 0 || N2 <= 0 || W <= 0) return CC_ERR_INVALID;
    if (metric != CC_METRIC_EUCLIDEAN && metric != CC_METRIC_COSINE) return CC_ERR_UNSUPPORTED;
    if (!(p > 0.f)) return CC_ERR_INVALID;
    if (self_nearest && N2 > N1) return CC_ERR_INVALID;          // the reference indexes dis[..., j, j] for j < N2
    if (all_negative && (!ws || ws_bytes < sizeof(unsigned))) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned* gmax = static_cast<unsigned*>(ws);
    if (all_negative && hipMemsetAsync(gmax, 0, sizeof(unsigned), st) != hipSuccess) return CC_ERR_HIP;
    const dim3 grid((N2 + 15) / 16, (N1 + 15) / 16, P);
    hipLaunchKernelGGL(cross_dist_kernel, grid, dim3(256), 0, st, x1, x2, N1, N2, W, metric, p, dist,
                       all_negative ? gmax : nullptr);
    CC_LAUNCH_CHECK();
    if (all_negative || self_nearest) {
        const int64_t total = (int64_t)P * N1 * N2;
        const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(cross_shift_kernel, dim3(blocks), dim3(256), 0, st, dist, P, N1, N2, all_negative ? gmax : nullptr,
                           self_nearest);
        CC_LAUNCH_CHECK();
    }
    return CC_OK;
}

int cc_kmedoids_from_dist_f32(const float* dist, const float* norms, int32_t P, int32_t N, int32_t K,
                              int32_t iter_limit, int32_t id_sort, int64_t* medoids, int64_t* assign, int32_t* iters,
                              void* ws, size_t ws_bytes, void* stream) {
    (void)ws; (void)ws_bytes;
    if (!dist || !norms || !medoids || P <= 0 || N <= 0 || K <= 0 || K > N || iter_limit < 0) return CC_ERR_INVALID;
    if (N > SEL_MAX_N) return CC_ERR_UNSUPPORTED;
    return run_select(dist, nullptr, norms, nullptr, 0, 1, 0, P, N, K, iter_limit, id_sort,
                      reinterpret_cast<long long*>(medoids), reinterpret_cast<long long*>(assign), iters,
                      static_cast<hipStream_t>(stream));
}

static int batch_kmedoids_impl(const float* x, const cc_token_layout* lay, int32_t W, int32_t K, int32_t metric,
                               float norm_p, float threshold, int32_t iter_limit, int32_t id_sort, int32_t split_size,
                               int32_t pre_norm, int64_t* medoids, int64_t* assign, int32_t* iters, void* ws,
                               size_t ws_bytes, void* stream, const GatherDesc* gather) {
    // Stop test: every problem iterates to its fixed point (medoids unchanged), which gives the final state of the
    // reference's chunk-mean test (fast_kmeans.py:85-88) whenever the threshold is below the distance between any two
    // distinct tokens (threshold <= CC_KMEDOIDS_MAX_THRESHOLD: one launch).  A looser threshold stops the reference earlier,
    // and at a point that depends on all problems of the split chunk: that case runs the literal loop - one iteration of
    // every problem per launch, the chunk's center_shift in ATen's own summation order after each, chunks that passed the
    // test passing through untouched (2 * iter_limit + 2 launches; not a hot path: no shipped script sets such a threshold).
    const bool literal_stop = !(threshold <= CC_KMEDOIDS_MAX_THRESHOLD);
    if (threshold != threshold) return CC_ERR_INVALID;
    if (!x || !medoids || !layout_ok(lay, W)) return CC_ERR_INVALID;
    const int P = lay->B * lay->S, N = lay->fd * lay->n;
    if (K <= 0 || K > N || iter_limit < 0) return CC_ERR_INVALID;
    if (metric != CC_METRIC_EUCLIDEAN && metric != CC_METRIC_COSINE) return CC_ERR_UNSUPPORTED;
    if (!p_supported(metric, norm_p)) return CC_ERR_UNSUPPORTED;
    if (N > SEL_MAX_N) return CC_ERR_UNSUPPORTED;
    if (split_size <= 0 || split_size > P) split_size = P;
    ClusterWs c = carve(ws, P, N, W, pre_norm, N);
    if (!ws || ws_bytes < c.total) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = run_distance(x, *lay, W, metric, norm_p, split_size, pre_norm, c, st);
    if (rc != CC_OK) return rc;
    long long* med_out = reinterpret_cast<long long*>(medoids);
    long long* asg_out = reinterpret_cast<long long*>(assign);
    if (!literal_stop || iter_limit == 0)
        return run_select(c.draw, c.draw, c.nrm, c.tilemax, c.slots_pp, split_size, 1, P, N, K, iter_limit, id_sort, med_out,
                          asg_out, iters, st, gather);
    if (split_size > 1024 || W >= 8192 || K >= 8192) return CC_ERR_UNSUPPORTED;      // (center_shift_kernel's per-chunk table / tree depth)
    const float* xs = pre_norm ? c.xn : x;                         // the tokens the reference's loop sees (fast_kmeans.py:21-22)
    const cc_token_layout lays = pre_norm ? contiguous_layout(P, N, W) : *lay;
    const int chunks = (P + split_size - 1) / split_size;
    if (hipMemsetAsync(c.done, 0, (size_t)chunks * sizeof(int), st) != hipSuccess) return CC_ERR_HIP;
    long long* cur = c.med;                                        // medoids entering a step / leaving it
    long long* nxt = c.med2;
    // KKZ init only (iter_limit 0, no sort): cur = the initial medoids, iters = 0
    rc = run_select(c.draw, c.draw, c.nrm, c.tilemax, c.slots_pp, split_size, 1, P, N, K, 0, 0, cur, nullptr, iters, st, nullptr);
    if (rc != CC_OK) return rc;
    float* rows_ws = c.sqn;                                        // [P, N] floats >= [P, K]: free once the distances exist
    for (int it = 0; it < iter_limit; ++it) {
        SelStep stp{cur, c.done, 1};
        rc = run_select(c.draw, c.draw, c.nrm, c.tilemax, c.slots_pp, split_size, 1, P, N, K, 1, 0, nxt, asg_out, iters, st,
                        nullptr, &stp);
        if (rc != CC_OK) return rc;
        hipLaunchKernelGGL(center_shift_kernel, dim3(chunks), dim3(256), 0, st, xs, lays, W, P, K, split_size, nxt, cur, threshold,
                           c.done, rows_ws);
        CC_LAUNCH_CHECK();
        long long* t = cur; cur = nxt; nxt = t;
    }
    if (id_sort || gather) {                                       // fast_kmeans.py:90-94 (+ the output rows)
        SelStep stp{cur, nullptr, 0};
        return run_select(c.draw, c.draw, c.nrm, c.tilemax, c.slots_pp, split_size, 1, P, N, K, 0, id_sort, med_out,
                          id_sort ? asg_out : nullptr, nullptr, st, gather, &stp);
    }
    if (hipMemcpyAsync(med_out, cur, (size_t)P * K * sizeof(long long), hipMemcpyDeviceToDevice, st) != hipSuccess) return CC_ERR_HIP;
    return CC_OK;
}

int cc_batch_kmedoids_f32(const float* x, const cc_token_layout* lay, int32_t W, int32_t K, int32_t metric,
                          float norm_p, float threshold, int32_t iter_limit, int32_t id_sort, int32_t split_size,
                          int32_t pre_norm, int64_t* medoids, int64_t* assign, int32_t* iters, void* ws,
                          size_t ws_bytes, void* stream) {
    return batch_kmedoids_impl(x, lay, W, K, metric, norm_p, threshold, iter_limit, id_sort, split_size, pre_norm, medoids,
                               assign, iters, ws, ws_bytes, stream, nullptr);
}

// *_rows: the public entry plus the by-products the fused forward wants from the same launch - row_h16 [rows][W] fp16
// copy of the output rows and row_stats [rows][2] their (sum, sum of squares); output must be dense ([seg][1+K][W]).
static bool rows_layout_ok(const _Float16* row_h16, const float* row_stats, int W, int Lout, int64_t out_tok, int64_t out_frame) {
    return !row_h16 || (row_stats && out_tok == W && out_frame == (int64_t)Lout * W);
}

int cc_token_gather_rows(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                         int32_t T_new, int32_t n, int32_t W, int32_t K, const int64_t* medoids, float* out,
                         int64_t out_tok_stride, int64_t out_frame_stride, _Float16* row_h16, float* row_stats,
                         float* row_shift, void* stream) {
    if (!x || !out || !medoids || B <= 0 || T <= 0 || T_new <= 0 || n <= 0 || W <= 0 || K <= 0) return CC_ERR_INVALID;
    if (!rows_layout_ok(row_h16, row_stats, W, 1 + K, out_tok_stride, out_frame_stride)) return CC_ERR_INVALID;
    if ((T % T_new) || (W & 3) || ((in_tok_stride | in_frame_stride | out_tok_stride | out_frame_stride) & 3))
        return CC_ERR_INVALID;
    launch_reduce_tokens( static_cast<hipStream_t>(stream), x,
                       in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, 0, reinterpret_cast<const long long*>(medoids), K,
                       (const long long*)nullptr, (const float*)nullptr, (const float*)nullptr, out, out_tok_stride,
                       out_frame_stride, row_h16, row_stats, row_shift);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_token_gather_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                        int32_t T_new, int32_t n, int32_t W, int32_t K, const int64_t* medoids, float* out,
                        int64_t out_tok_stride, int64_t out_frame_stride, void* stream) {
    return cc_token_gather_rows(x, in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, medoids, out, out_tok_stride,
                                out_frame_stride, nullptr, nullptr, nullptr, stream);
}

int cc_token_aggregate_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                           int32_t T_new, int32_t n, int32_t W, int32_t K, const int64_t* assign,
                           const cc_cluster_variant* var, float* out, int64_t out_tok_stride, int64_t out_frame_stride,
                           void* stream) {
    if (!x || !out || !assign || B <= 0 || T <= 0 || T_new <= 0 || n <= 0 || W <= 0 || K <= 0) return CC_ERR_INVALID;
    _Float16* const row_h16 = nullptr;
    float* const row_stats = nullptr;
    float* const row_shift = nullptr;
    if ((T % T_new) || (W & 3) || ((in_tok_stride | in_frame_stride | out_tok_stride | out_frame_stride) & 3))
        return CC_ERR_INVALID;
    launch_reduce_tokens( static_cast<hipStream_t>(stream), x,
                       in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, 1, (const long long*)nullptr, 0,
                       reinterpret_cast<const long long*>(assign), var ? var->cluster_embed : nullptr,
                       var ? var->cls_multiplier : nullptr, out, out_tok_stride, out_frame_stride, row_h16, row_stats,
                       row_shift);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_token_apply_selection_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                                 int32_t T_new, int32_t n, int32_t W, int32_t K, const cc_cluster_variant* var,
                                 const int64_t* medoids, const int64_t* assign, float* out, int64_t out_tok_stride,
                                 int64_t out_frame_stride, void* stream) {
    if (!x || !out || B <= 0 || T <= 0 || T_new <= 0 || n <= 0 || W <= 0 || K <= 0) return CC_ERR_INVALID;
    if ((T % T_new) || (W & 3) || ((in_tok_stride | in_frame_stride | out_tok_stride | out_frame_stride) & 3))
        return CC_ERR_INVALID;
    const bool mean = var && var->aggregation == CC_AGGREGATE_MEAN;
    if (mean ? !assign : !medoids) return CC_ERR_INVALID;
    _Float16* const row_h16 = nullptr;
    float* const row_stats = nullptr;
    float* const row_shift = nullptr;
    launch_reduce_tokens(
                       static_cast<hipStream_t>(stream), x, in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, mean ? 1 : 0,
                       mean ? (const long long*)nullptr : reinterpret_cast<const long long*>(medoids), mean ? 0 : K,
                       mean ? reinterpret_cast<const long long*>(assign) : (const long long*)nullptr,
                       var ? var->cluster_embed : nullptr, var ? var->cls_multiplier : nullptr, out, out_tok_stride,
                       out_frame_stride, row_h16, row_stats, row_shift);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_token_cluster_variant_rows(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                                  int32_t T_new, int32_t n, int32_t W, int32_t K, int32_t metric, float norm_p,
                                  float threshold, int32_t iter_limit, int32_t split_size, int32_t pre_norm,
                                  const cc_cluster_variant* var, float* out, int64_t out_tok_stride,
                                  int64_t out_frame_stride, int64_t* medoids, int64_t* assign, int32_t* iters, void* ws,
                                  size_t ws_bytes, _Float16* row_h16, float* row_stats, float* row_shift, void* stream) {
    if (!x || !out || !var || B <= 0 || T <= 0 || T_new <= 0 || n <= 0 || W <= 0) return CC_ERR_INVALID;
    if (!rows_layout_ok(row_h16, row_stats, W, 1 + (var->algorithm == CC_CLUSTER_POOLING ? n : K), out_tok_stride,
                        out_frame_stride))
        return CC_ERR_INVALID;
    if (T % T_new) return CC_ERR_INVALID;
    if ((W & 3) || ((in_tok_stride | in_frame_stride | out_tok_stride | out_frame_stride) & 3)) return CC_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int fd = T / T_new;
    if (var->algorithm == CC_CLUSTER_POOLING) {                 // no selection: every token = mean over the segment's frames
        launch_reduce_tokens( st, x, in_tok_stride, in_frame_stride,
                           B, T, T_new, n, W, n, 2, (const long long*)nullptr, 0, (const long long*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, out, out_tok_stride, out_frame_stride, row_h16, row_stats,
                           row_shift);
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    if (var->algorithm == CC_CLUSTER_SPARSE_SAMPLING) {         // fixed ids shared by every problem, then gather + CLS mean
        if (!var->fixed_ids || K <= 0) return CC_ERR_INVALID;
        launch_reduce_tokens( st, x, in_tok_stride, in_frame_stride,
                           B, T, T_new, n, W, K, 0, reinterpret_cast<const long long*>(var->fixed_ids), 0,
                           (const long long*)nullptr, (const float*)nullptr, (const float*)nullptr, out, out_tok_stride,
                           out_frame_stride, row_h16, row_stats, row_shift);
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    const bool spectral = var->algorithm == CC_CLUSTER_SPECTRAL;
    if ((var->algorithm != CC_CLUSTER_KMEDOIDS && !spectral) || K <= 0) return CC_ERR_INVALID;
    if (var->aggregation != CC_AGGREGATE_MEDOID && var->aggregation != CC_AGGREGATE_MEAN) return CC_ERR_INVALID;
    cc_token_layout lay;
    lay.B = B; lay.S = T_new; lay.fd = fd; lay.n = n;
    lay.stride_b = (int64_t)T * in_frame_stride;
    lay.stride_s = (int64_t)fd * in_frame_stride;
    lay.stride_f = in_frame_stride;
    lay.stride_i = in_tok_stride;
    const int P = B * T_new, N = fd * n;
    if (K > N) return CC_ERR_INVALID;
    ClusterWs c = carve(ws, P, N, W, pre_norm, N);
    if (!ws || ws_bytes < c.total) return CC_ERR_WORKSPACE;
    int64_t* med = medoids ? medoids : reinterpret_cast<int64_t*>(c.med);
    const bool mean = var->aggregation == CC_AGGREGATE_MEAN;
    int64_t* asg = assign ? assign : (mean ? reinterpret_cast<int64_t*>(c.asg) : nullptr);
    int rc;
    if (spectral) {
        // graph Laplacian of the segment's tokens -> K trailing eigenvectors -> k-medoids on their normalised rows
        // (spectral.py:42-73; a single chunk unless split_size > 1 and P > split_size, :64-72)
        SpectralWs sw = carve_spectral(static_cast<char*>(ws) + c.total, P, N, K);
        if (ws_bytes < c.total + sw.total) return CC_ERR_WORKSPACE;
        if (metric != CC_METRIC_EUCLIDEAN && metric != CC_METRIC_COSINE) return CC_ERR_INVALID;
        rc = cc_spectral_graph_laplacian_f32(x + in_tok_stride, &lay, W, var->spectral_sigma, var->spectral_graph_mode,
                                             var->spectral_knn_k, 0, var->spectral_graph, sw.lap, nullptr, nullptr, ws,
                                             c.total, stream);
        if (rc != CC_OK) return rc;
        rc = cc_spectral_embedding_f32(sw.lap, P, N, K, var->spectral_correct_sign, sw.q, sw.k4, nullptr, nullptr, sw.eig,
                                       sw.eig_bytes, stream);
        if (rc != CC_OK) return rc;
        cc_token_layout ql;
        ql.B = P; ql.S = 1; ql.fd = 1; ql.n = N;
        ql.stride_b = (int64_t)N * sw.k4; ql.stride_s = 0; ql.stride_f = 0; ql.stride_i = sw.k4;
        int64_t* asg_q = asg ? asg : reinterpret_cast<int64_t*>(c.asg);
        rc = cc_batch_kmedoids_f32(sw.q, &ql, sw.k4, K, metric, norm_p, threshold, iter_limit, 1,
                                   (split_size > 1 && P > split_size) ? split_size : P, 1, med, asg_q, iters, sw.km, sw.km_bytes,
                                   stream);
    } else {
        // the shipped variant: the selection kernel writes the output rows itself (K3 folded into K2's tail)
        const bool fold = !mean && (W & 31) == 0;
        const GatherDesc gd = gather_desc(x, in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, 0,
                                          reinterpret_cast<const long long*>(med), K, (const long long*)nullptr,
                                          var->cluster_embed, var->cls_multiplier, out, out_tok_stride, out_frame_stride,
                                          row_h16, row_stats, row_shift);
        rc = batch_kmedoids_impl(x + in_tok_stride, &lay, W, K, metric, norm_p, threshold, iter_limit, 1, split_size,
                                 pre_norm, med, asg, iters, ws, ws_bytes, stream, fold ? &gd : nullptr);
        if (rc == CC_OK && fold) return CC_OK;
    }
    if (rc != CC_OK) return rc;
    launch_reduce_tokens( st, x, in_tok_stride, in_frame_stride, B,
                       T, T_new, n, W, K, mean ? 1 : 0, reinterpret_cast<const long long*>(med), K,
                       reinterpret_cast<const long long*>(asg), var->cluster_embed, var->cls_multiplier, out,
                       out_tok_stride, out_frame_stride, row_h16, row_stats, row_shift);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_token_cluster_variant_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                                 int32_t T_new, int32_t n, int32_t W, int32_t K, int32_t metric, float norm_p,
                                 float threshold, int32_t iter_limit, int32_t split_size, int32_t pre_norm,
                                 const cc_cluster_variant* var, float* out, int64_t out_tok_stride,
                                 int64_t out_frame_stride, int64_t* medoids, int64_t* assign, int32_t* iters, void* ws,
                                 size_t ws_bytes, void* stream) {
    return cc_token_cluster_variant_rows(x, in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, metric, norm_p, threshold,
                                         iter_limit, split_size, pre_norm, var, out, out_tok_stride, out_frame_stride,
                                         medoids, assign, iters, ws, ws_bytes, nullptr, nullptr, nullptr, stream);
}

int cc_token_cluster_f32(const float* x, int64_t in_tok_stride, int64_t in_frame_stride, int32_t B, int32_t T,
                         int32_t T_new, int32_t n, int32_t W, int32_t K, int32_t metric, float norm_p,
                         float threshold, int32_t iter_limit, int32_t split_size, int32_t pre_norm, float* out,
                         int64_t out_tok_stride, int64_t out_frame_stride, int64_t* medoids, int64_t* assign,
                         int32_t* iters, void* ws, size_t ws_bytes, void* stream) {
    cc_cluster_variant var{};
    var.algorithm = CC_CLUSTER_KMEDOIDS;
    var.aggregation = CC_AGGREGATE_MEDOID;
    return cc_token_cluster_variant_f32(x, in_tok_stride, in_frame_stride, B, T, T_new, n, W, K, metric, norm_p, threshold,
                                        iter_limit, split_size, pre_norm, &var, out, out_tok_stride, out_frame_stride,
                                        medoids, assign, iters, ws, ws_bytes, stream);
}

}  // extern "C"

// ============================================================================ N4 (training support): backward of K3
// Gradient of TokenClusterInter.forward with respect to its input and its parameters, for a fixed selection (the medoid
// ids / assignment are piecewise constant in x: the reference runs them under no_grad, fast_kmeans.py:13,44).
//   medoid gather (cluster.py:289)      gx[token medoid_k] = g[1 + k]          (ids are distinct: every row written once)
//   cluster means (:291-301)            gx[j] = g[1 + assign_j] / |cluster(assign_j)|
//   cluster_embed add (:304-305)        g_embed[k] = sum over segments of g[1 + k]
//   CLS = mean over the segment's frames of (cls * cls_multiplier) (:244-245,307-308)
//                                       gx[cls of frame t] = g[0] / fd (* m_t);  g_m[t] = sum_{b,w} g[0] / fd * cls
//   pooling (:319-324)                  gx[every token] = g[same token] / fd
// One wave per INPUT row: each row of gx is written exactly once (zeros included) - no memset, no atomics, deterministic.
__global__ __launch_bounds__(256) void token_grad_kernel(const float* __restrict__ g, int64_t g_tok, int64_t g_frame, int B,
                                                         int T, int T_new, int n, int W, int K, int mode,
                                                         const long long* __restrict__ ids, int id_stride,
                                                         const long long* __restrict__ assign,
                                                         const float* __restrict__ cls_mult, float* __restrict__ gx,
                                                         int64_t gx_tok, int64_t gx_frame) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T * (1 + n)) return;
    const int frame = row / (1 + n), l = row - frame * (1 + n);
    const int b = frame / T, t = frame - b * T;
    const int fd = T / T_new, N = fd * n;
    const int sgm = t / fd, f = t - sgm * fd;
    const int seg = b * T_new + sgm, p = sgm * B + b;
    const float* gseg = g + (int64_t)seg * g_frame;
    float* dst = gx + (int64_t)l * gx_tok + (int64_t)frame * gx_frame;
    const float den = (float)fd;
    if (l == 0 || mode == 2) {
        const float* src = gseg + (int64_t)l * g_tok;
        const float m = (cls_mult && l == 0 && mode != 2) ? cls_mult[t] : 1.f;
        for (int w = lane * 4; w < W; w += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + w);
            v.x = v.x / den; v.y = v.y / den; v.z = v.z / den; v.w = v.w / den;
            if (cls_mult && l == 0 && mode != 2) { v.x *= m; v.y *= m; v.z *= m; v.w *= m; }
            *reinterpret_cast<float4*>(dst + w) = v;
        }
        return;
    }
    const int j = f * n + (l - 1);
    if (mode == 1) {
        const long long* a = assign + (int64_t)p * N;
        const long long k = a[j];
        int cnt = 0;
        for (int jj = lane; jj < N; jj += 64) cnt += (a[jj] == k) ? 1 : 0;
        const float c = cc_wave_sum((float)cnt);                  // exact: counts < 2^24
        const float* src = gseg + (int64_t)(1 + k) * g_tok;
        for (int w = lane * 4; w < W; w += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + w);
            v.x = v.x / c; v.y = v.y / c; v.z = v.z / c; v.w = v.w / c;
            *reinterpret_cast<float4*>(dst + w) = v;
        }
        return;
    }
    // gather: every k with ids[k] == j contributes (k-medoids ids are distinct; fixed ids of 'sparse_sampling' may repeat)
    const long long* idp = ids + (int64_t)p * id_stride;
    for (int w0 = 0; w0 < W; w0 += 256) {
        const int w = w0 + lane * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k0 = 0; k0 < K; k0 += 64) {
            const int k = k0 + lane;
            unsigned long long hit = __ballot(k < K && idp[k] == (long long)j);
            while (hit) {
                const int kk = k0 + __builtin_ctzll(hit);
                hit &= hit - 1;
                if (w < W) {
                    const float4 v = *reinterpret_cast<const float4*>(gseg + (int64_t)(1 + kk) * g_tok + w);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            }
        }
        if (w < W) *reinterpret_cast<float4*>(dst + w) = acc;
    }
}

// Parameter gradients: rows [0, K) -> g_embed[k] (sum over the segments, ascending), rows [K, K + T) -> g_mult[t].
__global__ __launch_bounds__(256) void token_param_grad_kernel(const float* __restrict__ g, int64_t g_tok, int64_t g_frame,
                                                               const float* __restrict__ x, int64_t x_frame, int B, int T,
                                                               int T_new, int W, int K, float* __restrict__ g_embed,
                                                               float* __restrict__ g_mult) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= K + T) return;
    const int segs = B * T_new, fd = T / T_new;
    if (row < K) {
        if (!g_embed) return;
        for (int w = lane * 4; w < W; w += 256) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int sg = 0; sg < segs; ++sg) {
                const float4 v = *reinterpret_cast<const float4*>(g + (int64_t)(1 + row) * g_tok + (int64_t)sg * g_frame + w);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            *reinterpret_cast<float4*>(g_embed + (int64_t)row * W + w) = acc;
        }
        return;
    }
    if (!g_mult) return;
    const int t = row - K, sgm = t / fd;
    const float den = (float)fd;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* gs = g + (int64_t)(b * T_new + sgm) * g_frame;
        const float* xs = x + (int64_t)(b * T + t) * x_frame;
        for (int w = lane; w < W; w += 64) acc += (gs[w] / den) * xs[w];
    }
    acc = cc_wave_sum(acc);
    if (lane == 0) g_mult[t] = acc;
}

extern "C" int cc_token_cluster_backward_f32(const float* grad_out, int64_t go_tok_stride, int64_t go_frame_stride,
                                             int32_t B, int32_t T, int32_t T_new, int32_t n, int32_t W, int32_t K,
                                             const cc_cluster_variant* var, const int64_t* medoids,
                                             const int64_t* assign, const float* x, int64_t x_tok_stride,
                                             int64_t x_frame_stride, float* grad_x, int64_t gx_tok_stride,
                                             int64_t gx_frame_stride, float* grad_cluster_embed, float* grad_cls_mult,
                                             void* stream) {
    (void)x_tok_stride;
    if (!grad_out || !grad_x || !var || B <= 0 || T <= 0 || T_new <= 0 || n <= 0 || W <= 0 || K <= 0) return CC_ERR_INVALID;
    if ((T % T_new) || (W & 3)) return CC_ERR_INVALID;
    const int fd = T / T_new, N = fd * n;
    int mode = 0, id_stride = K;
    const long long* ids = reinterpret_cast<const long long*>(medoids);
    if (var->algorithm == CC_CLUSTER_POOLING) {
        if (K != n) return CC_ERR_INVALID;
        mode = 2;
    } else if (var->algorithm == CC_CLUSTER_SPARSE_SAMPLING) {
        if (!var->fixed_ids) return CC_ERR_INVALID;
        ids = reinterpret_cast<const long long*>(var->fixed_ids);
        id_stride = 0;
    } else if (var->algorithm == CC_CLUSTER_KMEDOIDS) {
        if (K > N) return CC_ERR_INVALID;
        if (var->aggregation == CC_AGGREGATE_MEAN) {
            if (!assign) return CC_ERR_INVALID;
            mode = 1;
        } else if (!medoids) {
            return CC_ERR_INVALID;
        }
    } else {
        return CC_ERR_UNSUPPORTED;
    }
    if (grad_cls_mult && !x) return CC_ERR_INVALID;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int rows = B * T * (1 + n);
    hipLaunchKernelGGL(token_grad_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, grad_out, go_tok_stride, go_frame_stride, B,
                       T, T_new, n, W, K, mode, ids, id_stride, reinterpret_cast<const long long*>(assign),
                       mode == 2 ? nullptr : var->cls_multiplier, grad_x, gx_tok_stride, gx_frame_stride);
    if (hipGetLastError() != hipSuccess) return CC_ERR_HIP;
    if (grad_cluster_embed || grad_cls_mult) {
        if (mode == 2) return CC_ERR_INVALID;
        hipLaunchKernelGGL(token_param_grad_kernel, dim3((K + T + 3) / 4), dim3(256), 0, st, grad_out, go_tok_stride,
                           go_frame_stride, x, x_frame_stride, B, T, T_new, W, K, grad_cluster_embed, grad_cls_mult);
        if (hipGetLastError() != hipSuccess) return CC_ERR_HIP;
    }
    return CC_OK;
}

// ============================================================================ N4 (spectral clustering)
// modules/cluster/spectral.py:17-165:
//   constructW ('HeatKernel', optional spatial-temporal mask)   W = exp(-|x_i - x_j|^2 / (2 sigma^2)) [* graph]   (:79-107)
//   normalised Laplacian                                         L_sym = D^-1/2 (D - W) D^-1/2, D = diag(W 1)   (:44-52)
//   batch_sign_flip_rasmus_bro                                   U[:, k] *= sign(sum_j sign(s_k v_kj) (s_k v_kj)^2) (:110-137)
// The squared distances come from the Gram kernel above (raw, unshifted D of METRIC 2 = squared L2, no clamp, no sqrt:
// batched_cdist_l2, cluster_utils.py:121-133).

// W[p][i][j] = exp(-d2 / (2 sigma^2)) (* graph[i][j]); deg[p][i] = sum_j W[p][i][j].  One wave per row, in place on d2.
__global__ __launch_bounds__(256) void heat_kernel_rows_kernel(float* __restrict__ d2w, const unsigned char* __restrict__ graph,
                                                               float* __restrict__ deg, int P, int N, float inv_two_sigma2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P * N) return;
    const int i = row % N;
    float* r = d2w + (int64_t)row * N;
    float s = 0.f;
    for (int j = lane; j < N; j += 64) {
        float w = expf(-1.0f * r[j] * inv_two_sigma2);
        if (graph) w = graph[(int64_t)i * N + j] ? w : 0.f;
        r[j] = w;
        s += w;
    }
    s = cc_wave_sum(s);
    if (lane == 0) deg[row] = s;
}

// L[p][i][j] = ((i == j ? deg_i : 0) - W_ij) * deg_i^-1/2 * deg_j^-1/2   (the two bmm's with diagonal matrices, :51)
__global__ __launch_bounds__(256) void sym_laplacian_kernel(const float* __restrict__ w, const float* __restrict__ deg,
                                                            float* __restrict__ L, int P, int N) {
    const int64_t total = (int64_t)P * N * N;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int p = (int)(idx / ((int64_t)N * N));
        const int rem = (int)(idx - (int64_t)p * N * N);
        const int i = rem / N, j = rem - i * N;
        const float di = deg[(int64_t)p * N + i], dj = deg[(int64_t)p * N + j];
        const float l = (i == j ? di : 0.f) - w[idx];
        L[idx] = (powf(di, -0.5f) * l) * powf(dj, -0.5f);
    }
}

// constructW 'KNN' (spectral.py:89-100): kth[p][i] = the knn_k-th largest entry of row i of the heat-kernel affinity
// (torch.topk(W, knn_k)[..., -1]: duplicates count).  One wave per row: repeatedly the largest value below the previous
// one, with its multiplicity, until knn_k entries are covered.
__global__ __launch_bounds__(256) void knn_threshold_rows_kernel(const float* __restrict__ w, float* __restrict__ kth, int P,
                                                                 int N, int knn_k) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P * N) return;
    const float* r = w + (int64_t)row * N;
    float bound = __builtin_inff();
    int covered = 0;
    float cur = 0.f;
    while (covered < knn_k) {
        float m = -__builtin_inff();
        for (int j = lane; j < N; j += 64) {
            const float v = r[j];
            if (v < bound) m = fmaxf(m, v);
        }
        m = cc_wave_max(m);
        if (!(m > -__builtin_inff())) break;                     // fewer than knn_k entries in the row
        int c = 0;
        for (int j = lane; j < N; j += 64) c += (r[j] == m) ? 1 : 0;
        covered += (int)cc_wave_sum((float)c);
        cur = m;
        bound = m;
    }
    if (lane == 0) kth[row] = cur;
}

// W_ij *= (W_ij >= kth_i) or/and (W_ji >= kth_j) (spectral.py:93-99), then the optional spatial-temporal mask (:104-105)
// and the degrees.  One wave per row; W is symmetric, so W_ji is read as W_ij.
__global__ __launch_bounds__(256) void knn_mask_rows_kernel(float* __restrict__ w, const float* __restrict__ kth,
                                                            const unsigned char* __restrict__ graph, float* __restrict__ deg,
                                                            int P, int N, int mutual) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= P * N) return;
    const int p = row / N, i = row - p * N;
    float* r = w + (int64_t)row * N;
    const float ki = kth[row];
    float s = 0.f;
    for (int j = lane; j < N; j += 64) {
        const float v = r[j];
        const bool a = v >= ki, b = v >= kth[(int64_t)p * N + j];
        float o = (mutual ? (a && b) : (a || b)) ? v : 0.f;
        if (graph) o = graph[(int64_t)i * N + j] ? o : 0.f;
        r[j] = o;
        s += o;
    }
    s = cc_wave_sum(s);
    if (lane == 0) deg[row] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// The decomposition step of batch_spectral_clustering (spectral.py:54-61): the K eigenvectors of L_sym with the smallest
// eigenvalues (the reference takes the trailing K left singular vectors of torch.linalg.svd; L_sym is symmetric positive
// semi-definite, so they are those eigenvectors up to sign / a rotation inside degenerate eigenspaces).
// One-sided (Hestenes) Jacobi on G = 2I - L_sym (eigenvalues mu = 2 - lambda in [0, 2]: the wanted vectors belong to the
// LARGEST mu, where the one-sided method is accurate to working precision).  One workgroup per problem, G held row-major in
// LDS (N <= 201) or in a global scratch (larger N); a "vector" is a contiguous row.  Every round rotates N/2 disjoint row
// pairs (round-robin tournament order): alpha = |x|^2, beta = |y|^2, gamma = x.y by one pass + DPP sums, rotation by the
// smaller angle, rows rewritten in place.  At convergence row_j = mu_j u_j^T.
// Output Q[p][i][c], c = K - 1 - rank(mu_j descending) - the reference's column order (singular values descending, last K
// columns) - with the sign of batch_sign_flip_rasmus_bro applied when correct_sign (for a symmetric matrix the flip only
// depends on the vector itself); eigenvalues lambda = 2 - mu in the same order.
constexpr int EIG_WAVES = 16;

// MAXE = ceil(N / 16): a row pair is owned by one 16-lane DPP row (lane g holds elements g, g + 16, ...), four pairs per
// wave, so the three dot products reduce inside the DPP row (no cross-row step) and the rotation arithmetic is issued once
// for four pairs.
template <int MAXE, bool IN_LDS>
__global__ __launch_bounds__(64 * EIG_WAVES) void sym_eig_jacobi_kernel(const float* __restrict__ Lsym, float* __restrict__ work,
                                                                        float* __restrict__ Q, float* __restrict__ evals,
                                                                        int* __restrict__ sweeps_out, int N, int K, int ldq,
                                                                        int correct_sign, int max_sweeps, float tol) {
    extern __shared__ __align__(16) unsigned char eig_smem[];
    const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* A = IN_LDS ? reinterpret_cast<float*>(eig_smem) : work + (int64_t)p * N * N;
    float* mu = reinterpret_cast<float*>(eig_smem) + (IN_LDS ? (size_t)N * N : 0);
    int* rank = reinterpret_cast<int*>(mu + N);
    unsigned* flag = reinterpret_cast<unsigned*>(rank + N);
    const float* Lp = Lsym + (int64_t)p * N * N;
    for (int idx = tid; idx < N * N; idx += 64 * EIG_WAVES) {
        const int i = idx / N, j = idx - i * N;
        A[idx] = (i == j ? 2.f : 0.f) - Lp[idx];
    }
    const int m = N + (N & 1);                                   // even number of players (the odd one sits out)
    const int grp = lane >> 4, gl = lane & 15;
    int sweeps = 0;
    for (; sweeps < max_sweeps; ++sweeps) {
        if (tid == 0) *flag = 0u;
        __syncthreads();
        float off = 0.f;
        for (int r = 0; r < m - 1; ++r) {
            for (int k = wave * 4 + grp; k < m / 2 + 3; k += 4 * EIG_WAVES) {      // (+3: the groups of a wave stay together)
                int i = r + k;                                   // (r + k) mod (m - 1), both < m - 1 when k is in range
                if (i >= m - 1) i -= m - 1;
                int j = r - k;
                if (j < 0) j += m - 1;
                if (k == 0) j = m - 1;
                const bool on = k < m / 2 && i < N && j < N;
                if (!on) { i = 0; j = 0; }
                if (i > j) { const int t = i; i = j; j = t; }
                float* xi = A + (int64_t)i * N + gl;
                float* xj = A + (int64_t)j * N + gl;
                float x[MAXE], y[MAXE];
                float al = 0.f, be = 0.f, ga = 0.f;
#pragma unroll
                for (int t = 0; t < MAXE; ++t) {
                    const bool in = gl + 16 * t < N;
                    x[t] = in ? xi[16 * t] : 0.f;
                    y[t] = in ? xj[16 * t] : 0.f;
                    al += x[t] * x[t];
                    be += y[t] * y[t];
                    ga += x[t] * y[t];
                }
                al += cc_dpp_f32<0xB1>(al); be += cc_dpp_f32<0xB1>(be); ga += cc_dpp_f32<0xB1>(ga);
                al += cc_dpp_f32<0x4E>(al); be += cc_dpp_f32<0x4E>(be); ga += cc_dpp_f32<0x4E>(ga);
                al += cc_dpp_f32<0x141>(al); be += cc_dpp_f32<0x141>(be); ga += cc_dpp_f32<0x141>(ga);
                al += cc_dpp_f32<0x140>(al); be += cc_dpp_f32<0x140>(be); ga += cc_dpp_f32<0x140>(ga);
                const float ab = al * be;
                const float rel = fabsf(ga) * __builtin_amdgcn_rsqf(fmaxf(ab, 1e-37f));     // |gamma| / sqrt(alpha beta)
                if (on && rel > tol) {
                    off = fmaxf(off, rel);
                    const float zeta = (be - al) * __builtin_amdgcn_rcpf(2.f * ga);
                    const float t = copysignf(1.f, zeta) * __builtin_amdgcn_rcpf(fabsf(zeta) + __builtin_amdgcn_sqrtf(1.f + zeta * zeta));
                    const float d = 1.f + t * t;
                    float c = __builtin_amdgcn_rsqf(d);
                    c = c * (1.5f - 0.5f * d * c * c);           // one Newton step: c^2 (1 + t^2) = 1 to rounding
                    const float sn = c * t;
#pragma unroll
                    for (int q = 0; q < MAXE; ++q) {
                        if (gl + 16 * q < N) {
                            xi[16 * q] = c * x[q] - sn * y[q];
                            xj[16 * q] = sn * x[q] + c * y[q];
                        }
                    }
                }
            }
            __syncthreads();
        }
        off = cc_wave_max(off);
        if (lane == 0 && off > 0.f) atomicMax(flag, __float_as_uint(off));       // (non-negative floats order as unsigned)
        __syncthreads();
        const float worst = __uint_as_float(*flag);
        __syncthreads();
        if (!(worst > tol)) { ++sweeps; break; }
    }
    if (tid == 0 && sweeps_out) sweeps_out[p] = sweeps;
    // mu_j = |row_j|
    for (int j = wave; j < N; j += EIG_WAVES) {
        float a = 0.f;
        for (int c = lane; c < N; c += 64) { const float v = A[(int64_t)j * N + c]; a += v * v; }
        a = cc_wave_sum_fast(a);
        if (lane == 0) mu[j] = sqrtf(a);
    }
    __syncthreads();
    for (int j = tid; j < N; j += 64 * EIG_WAVES) {
        const float mj = mu[j];
        int rk = 0;
        for (int q = 0; q < N; ++q) rk += (mu[q] > mj || (mu[q] == mj && q < j)) ? 1 : 0;
        rank[j] = rk;
    }
    __syncthreads();
    float* Qp = Q + (int64_t)p * N * ldq;
    for (int j = wave; j < N; j += EIG_WAVES) {
        const int rk = rank[j];
        if (rk >= K) continue;
        const int col = K - 1 - rk;
        const float inv = 1.f / mu[j];
        float sg = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float u = A[(int64_t)j * N + c] * inv;
            sg += (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)) * (u * u);
        }
        sg = cc_wave_sum_fast(sg);
        const float flip = correct_sign ? (sg > 0.f ? 1.f : (sg < 0.f ? -1.f : 0.f)) : 1.f;
        for (int c = lane; c < N; c += 64) Qp[(int64_t)c * ldq + col] = A[(int64_t)j * N + c] * inv * flip;
        if (lane == 0 && evals) evals[(int64_t)p * K + col] = 2.f - mu[j];
    }
}

// sign_left[p][k] = sum_j sign(S_k VT_kj) (S_k VT_kj)^2; U[p][:, k] *= sign(sign_left[p][k]).  One workgroup per (p, k).
__global__ __launch_bounds__(256) void svd_sign_flip_kernel(float* __restrict__ U, const float* __restrict__ S,
                                                            const float* __restrict__ VT, int M, int Kc, int Ncols) {
    __shared__ float red[4];
    const int p = blockIdx.y, k = blockIdx.x, tid = threadIdx.x;
    const float sk = S[(int64_t)p * Kc + k];
    const float* v = VT + ((int64_t)p * Kc + k) * Ncols;
    float acc = 0.f;
    for (int j = tid; j < Ncols; j += 256) {
        const float t = sk * v[j];
        const float sg = (t > 0.f) ? 1.f : ((t < 0.f) ? -1.f : 0.f);
        acc += sg * (t * t);
    }
    acc = cc_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    const float sg = (tot > 0.f) ? 1.f : ((tot < 0.f) ? -1.f : 0.f);
    float* u = U + (int64_t)p * M * Kc + k;
    for (int i = tid; i < M; i += 256) u[(int64_t)i * Kc] *= sg;
}

extern "C" {

int cc_spectral_laplacian_f32(const float* x, const cc_token_layout* lay, int32_t W, float sigma,
                              const uint8_t* graph, float* laplacian, float* affinity_out, float* degree_out, void* ws,
                              size_t ws_bytes, void* stream) {
    return cc_spectral_graph_laplacian_f32(x, lay, W, sigma, CC_GRAPH_HEAT_KERNEL, 0, 0, graph, laplacian, affinity_out,
                                           degree_out, ws, ws_bytes, stream);
}

int cc_spectral_graph_laplacian_f32(const float* x, const cc_token_layout* lay, int32_t W, float sigma, int32_t mode,
                                    int32_t knn_k, int32_t mutual, const uint8_t* graph, float* laplacian,
                                    float* affinity_out, float* degree_out, void* ws, size_t ws_bytes, void* stream) {
    if (!x || !laplacian || !layout_ok(lay, W) || !(sigma > 0.f)) return CC_ERR_INVALID;
    if (mode != CC_GRAPH_HEAT_KERNEL && mode != CC_GRAPH_KNN) return CC_ERR_UNSUPPORTED;
    if (mode == CC_GRAPH_KNN && (knn_k <= 0 || knn_k > lay->fd * lay->n)) return CC_ERR_INVALID;
    const int P = lay->B * lay->S, N = lay->fd * lay->n;
    ClusterWs c = carve(ws, P, N, W, 0, N);
    if (!ws || ws_bytes < c.total) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* wbuf = affinity_out ? affinity_out : c.draw;
    float* deg = degree_out ? degree_out : c.sqn;               // (sqn is free once the Gram kernel has run)
    ClusterWs g = c;
    g.draw = wbuf;
    g.sqn = c.nrm;                                              // keep the norm scratch apart from `deg`
    int rc = run_distance_sq(x, *lay, W, g, st);
    if (rc != CC_OK) return rc;
    const bool knn = mode == CC_GRAPH_KNN;
    hipLaunchKernelGGL(heat_kernel_rows_kernel, dim3((P * N + 3) / 4), dim3(256), 0, st, wbuf, knn ? nullptr : graph, deg, P, N,
                       1.0f / (2.0f * sigma * sigma));
    CC_LAUNCH_CHECK();
    if (knn) {                                                  // threshold per row, then mask (+ graph) and the degrees
        float* kth = c.nrm;
        hipLaunchKernelGGL(knn_threshold_rows_kernel, dim3((P * N + 3) / 4), dim3(256), 0, st, wbuf, kth, P, N, knn_k);
        CC_LAUNCH_CHECK();
        hipLaunchKernelGGL(knn_mask_rows_kernel, dim3((P * N + 3) / 4), dim3(256), 0, st, wbuf, kth, graph, deg, P, N, mutual);
        CC_LAUNCH_CHECK();
    }
    const int64_t total = (int64_t)P * N * N;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(sym_laplacian_kernel, dim3(blocks), dim3(256), 0, st, wbuf, deg, laplacian, P, N);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

size_t cc_spectral_embedding_workspace_bytes(int32_t P, int32_t N) {
    if (P <= 0 || N <= 0) return 0;
    if (N <= 196) return cc_sym_eig_tridiag_ws_bytes(P, N);      // the direct solver's band scratch (eig.hip)
    const size_t jac = N <= 201 ? 256 : cc_align_up((size_t)P * N * N * sizeof(float), 256);
    const size_t big = N <= 832 ? cc_sym_eig_tridiag_big_ws_bytes(P, N) : 0;     // matrix + fp64 vectors + bands in global memory
    return jac > big ? jac : big;
}

int cc_spectral_embedding_solver_f32(const float* laplacian, int32_t P, int32_t N, int32_t K, int32_t correct_sign,
                                     float* Q, int32_t ldq, float* eigenvalues, int32_t* sweeps_out, int32_t solver,
                                     void* ws, size_t ws_bytes, void* stream) {
    if (!laplacian || !Q || P <= 0 || N <= 1 || K <= 0 || K > N || ldq < K) return CC_ERR_INVALID;
    if (solver != CC_EIG_AUTO && solver != CC_EIG_JACOBI) return CC_ERR_INVALID;
    const bool g_force_jacobi = solver == CC_EIG_JACOBI;
    // (the Jacobi kernel stops at N = 640; 640 < N <= 832 - ViT-B/16 with four frames per segment, N = 784, K = 160 - exists
    // in the direct solver only)
    if (N > 640 && (g_force_jacobi || !cc_sym_eig_tridiag_big_supports(N, K))) return CC_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (ldq > K && hipMemsetAsync(Q, 0, (size_t)P * N * ldq * sizeof(float), st) != hipSuccess) return CC_ERR_HIP;
    if (!g_force_jacobi && (cc_sym_eig_tridiag_supports(N, K) || cc_sym_eig_tridiag_big_supports(N, K)))    // direct solver
        return cc_launch_sym_eig_tridiag(laplacian, P, N, K, correct_sign, Q, ldq, eigenvalues, sweeps_out, ws, ws_bytes, st);
    const bool in_lds = N <= 201;
    if (!in_lds && (!ws || ws_bytes < cc_spectral_embedding_workspace_bytes(P, N))) return CC_ERR_WORKSPACE;
    const size_t smem = (in_lds ? (size_t)N * N * 4 : 0) + (size_t)N * 8 + 64;
    const int max_sweeps = 30;
    // stop when every pair's |gamma| / sqrt(alpha beta) is below the rounding floor of a length-N fp32 dot product (a fixed
    // 1e-6 sits under that floor for N >~ 70: the loop could then run to max_sweeps without ever meeting it)
    const float tol = fmaxf(1e-6f, 0.25f * (float)N * 1.1920929e-7f);
#define EIG_LAUNCH(MAXR, INLDS)                                                                                        \
    do {                                                                                                               \
        auto kern = sym_eig_jacobi_kernel<MAXR, INLDS>;                                                                \
        if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return CC_ERR_HIP;                                                                                         \
        hipLaunchKernelGGL(kern, dim3(P), dim3(64 * EIG_WAVES), smem, st, laplacian, static_cast<float*>(ws), Q, eigenvalues, \
                           sweeps_out, N, K, ldq, correct_sign, max_sweeps, tol);                                     \
    } while (0)
    if (in_lds) {
        if (N <= 64) EIG_LAUNCH(4, true);
        else if (N <= 128) EIG_LAUNCH(8, true);
        else EIG_LAUNCH(13, true);
    } else if (N <= 256) {
        EIG_LAUNCH(16, false);
    } else if (N <= 400) {
        EIG_LAUNCH(25, false);
    } else {
        EIG_LAUNCH(40, false);
    }
#undef EIG_LAUNCH
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int cc_spectral_embedding_f32(const float* laplacian, int32_t P, int32_t N, int32_t K, int32_t correct_sign,
                              float* Q, int32_t ldq, float* eigenvalues, int32_t* sweeps_out, void* ws, size_t ws_bytes,
                              void* stream) {
    return cc_spectral_embedding_solver_f32(laplacian, P, N, K, correct_sign, Q, ldq, eigenvalues, sweeps_out, CC_EIG_AUTO,
                                            ws, ws_bytes, stream);
}

int cc_svd_sign_flip_f32(float* U, const float* S, const float* VT, int32_t P, int32_t M, int32_t K, int32_t N,
                         void* stream) {
    if (!U || !S || !VT || P <= 0 || M <= 0 || K <= 0 || N <= 0) return CC_ERR_INVALID;
    hipLaunchKernelGGL(svd_sign_flip_kernel, dim3(K, P), dim3(256), 0, static_cast<hipStream_t>(stream), U, S, VT, M, K, N);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

}  // extern "C"
