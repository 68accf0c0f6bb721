// Direct symmetric eigensolver for the decomposition step of batch_spectral_clustering (reference: modules/cluster/
// spectral.py:54-61 takes the K trailing left singular vectors of L_sym from torch.linalg.svd = the eigenvectors of the K
// smallest eigenvalues).  The one-sided Jacobi kernel in cluster.hip decomposes the whole N x N matrix (8-10 sweeps of N/2
// dependent rounds); this one does the work the K wanted pairs need, one workgroup of 16 waves per problem:
//
//   A  L_sym (symmetrised on load) -> LDS
//   B  Householder tridiagonalisation T = Q^T L Q in fp32, ONE pass over the trailing block per reflector (the rank-2
//      update of reflector k-1 and the matrix-vector product for reflector k fused), two barriers per step; the reflectors
//      stay in row k of the matrix and are then packed to the front of the LDS region, which frees the rest of it
//   C  eigenvalues of T in fp64: Sturm counts by the three-term recurrence (no division), 4 lanes per wanted eigenvalue,
//      5-section steps from the Gershgorin interval down to 1e-13 |T|
//   D  eigenvectors of T in fp64: inverse iteration, one lane per eigenvalue (Gaussian elimination with partial pivoting
//      on the shifted tridiagonal; the bands of U stream through a global scratch, lane-contiguous), and after every solve
//      a modified Gram-Schmidt pass over ALL K vectors in eigenvalue order (16 lanes per vector, vector in registers, the
//      pivot vector broadcast through LDS: one barrier per vector)
//   E  back-transformation z = H_0 ... H_{N-3} y in fp32 in the same lanes (reflectors read from LDS, no barrier)
//   F  the reference's column order (eigenvalue ascending -> column K-1-k), sign rule of batch_sign_flip_rasmus_bro
//
// Why C and D run in fp64 although T is only an fp32-accurate image of L: Gram-Schmidt amplifies whatever the solved
// vectors carry outside the wanted invariant subspace by the condition number of the solved block, and with shifts known to
// fp32 (1e-7 |T|) a cluster whose spacing is comparable to that (planted partitions: eigenvalue 0 of multiplicity K up to
// the coupling between the parts) gives blocks of condition 1e2 - 1e4, i.e. residuals of 1e-4 (measured: DESIGN.md,
// spectral decomposition).  As a matrix of exact numbers T has simple eigenvalues; with shifts accurate to 1e-13 |T| every solve is
// dominated by its own eigenvector, the block is orthogonal to rounding before it is orthogonalised, and the fp32 floor is
// what remains: residual |L q - lambda q| <= 4e-7, orthonormal to 1.3e-6, eigenvalues within 5e-7 of a float64 eigh over
// heat-kernel / KNN / planted-partition Laplacians (coupling 0 ... 1e-3, identical blocks) at N = 196, K = 49.
// Scope: the matrix in LDS next to 10 KB of vectors (N <= 196) and packed reflectors + K fp64 vectors in its place
// afterwards (K = 49 at N = 196), K <= 64; 196 < N <= 832 (K <= 192): sym_eig_tridiag_big_kernel below, the matrix in a global
// scratch; what is left (K > 49 at N = 196, K > 128) keeps the Jacobi kernel.
#include "cc_common.h"
#include "cc_kernels.h"

namespace {

constexpr int TD_WAVES = 16;
constexpr int TD_THREADS = 64 * TD_WAVES;
constexpr int TD_ITERS = 1;                 // inverse iterations: with fp64 shifts one reaches the fp32 floor (probe: 76 matrices)
constexpr int TD_SECTIONS = 19;             // 5-section steps: 5^19 = 2e13
constexpr double TD_EPS64 = 2.220446049250313e-16;

__device__ __forceinline__ float td_row16_sum(float v) {
    v += cc_dpp_f32<0xB1>(v);
    v += cc_dpp_f32<0x4E>(v);
    v += cc_dpp_f32<0x141>(v);
    v += cc_dpp_f32<0x140>(v);
    return v;
}
// x[l] + x[l^16] + x[l^32] + x[l^48] in every lane: the gfx950 row / half swaps instead of two ds_bpermute round trips
__device__ __forceinline__ float td_sum_rows(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
template <int CTRL>
__device__ __forceinline__ double td_dpp_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double td_row16_sum(double v) {
    v += td_dpp_f64<0xB1>(v);
    v += td_dpp_f64<0x4E>(v);
    v += td_dpp_f64<0x141>(v);
    v += td_dpp_f64<0x140>(v);
    return v;
}
__device__ __forceinline__ double td_row16_max(double v) {
    v = fmax(v, td_dpp_f64<0xB1>(v));
    v = fmax(v, td_dpp_f64<0x4E>(v));
    v = fmax(v, td_dpp_f64<0x141>(v));
    v = fmax(v, td_dpp_f64<0x140>(v));
    return v;
}

// Number of eigenvalues of T below x: sign changes of p_0 = 1, p_i = (d_i - x) p_{i-1} - e_{i-1}^2 p_{i-2}; ds[i] = (d_i,
// e_i^2 clamped away from 0 so that an exact zero of p cannot stick), readable up to index N + 15.  A zero counts as positive,
// which gives the same total as LAPACK's pivot replacement.  Per row: add, mul, fma and one v_alignbit that shifts the sign
// of p into a mask; the changes are counted per 8 rows (popcount of mask ^ mask >> 1), where p is also rescaled by its
// exponent.  Two register sets of 8 rows alternate so that the next rows are in flight while 8 are consumed.
#define TD_STURM_ROW(R)                                                       \
    {                                                                          \
        const double pn = ((R).x - x) * p1 - e2 * p0;                          \
        m = __builtin_amdgcn_alignbit(m, (unsigned)__double2hiint(pn), 31);    \
        p0 = p1; p1 = pn; e2 = (R).y;                                          \
    }
#define TD_STURM_CLOSE()                                                      \
    {                                                                          \
        c += __builtin_popcount((m ^ (m >> 1)) & 0xFFu);                       \
        const int ex = __builtin_amdgcn_frexp_exp(p1 != 0.0 ? p1 : p0);        \
        p1 = __builtin_amdgcn_ldexp(p1, -ex);                                  \
        p0 = __builtin_amdgcn_ldexp(p0, -ex);                                  \
    }
__device__ __forceinline__ int td_sturm(const double2* __restrict__ ds, int N, double x) {
    double p0 = 1.0, p1 = ds[0].x - x;
    unsigned m = (unsigned)__double2hiint(p1) >> 31;             // bit 0 = sign of the latest p; the sign of p_0 is 0
    int c = (int)m;
    double e2 = ds[0].y;
    double2 ra[8], rb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ra[u] = ds[1 + u];
    int i = 1;
    while (i + 16 <= N) {
#pragma unroll
        for (int u = 0; u < 8; ++u) rb[u] = ds[i + 8 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) TD_STURM_ROW(ra[u])
        TD_STURM_CLOSE()
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = ds[i + 16 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) TD_STURM_ROW(rb[u])
        TD_STURM_CLOSE()
        i += 16;
    }
    if (i + 8 <= N) {
#pragma unroll
        for (int u = 0; u < 8; ++u) rb[u] = ds[i + 8 + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) TD_STURM_ROW(ra[u])
        TD_STURM_CLOSE()
#pragma unroll
        for (int u = 0; u < 8; ++u) ra[u] = rb[u];
        i += 8;
    }
#pragma unroll
    for (int u = 0; u < 7; ++u) {                                // tail (< 8 rows): ra holds rows i, i + 1, ...
        if (i + u < N) {
            TD_STURM_ROW(ra[u])
            c += (int)((m ^ (m >> 1)) & 1u);
        }
    }
    return c;
}
#undef TD_STURM_ROW
#undef TD_STURM_CLOSE

__device__ __forceinline__ double td_start_value(unsigned i, unsigned k) {
    unsigned h = i * 0x9E3779B1u + k * 0x85EBCA77u + 0x165667B1u;
    h ^= h >> 15; h *= 0x2C1B3C6Du;
    h ^= h >> 12; h *= 0x297A2D39u;
    h ^= h >> 15;
    return (double)(h >> 8) * (1.0 / 8388608.0) - 1.0;
}

__device__ __forceinline__ int td_pack_offset(int k, int N) { return (k * (2 * N - 3 - k)) >> 1; }

// Back-transformation, reflectors kr whose first non-zero element f = kr + 1 lies in pair slot T0 (f in [32 T0, 32 T0 + 31]):
// slots below T0 are all-zero and skipped statically, slot T0 takes the (0 ... 0, 1, v ...) boundary, the rest is plain.
template <int MAXT, int T0>
__device__ __forceinline__ void td_back_range(float2 (&y)[MAXT], const float* __restrict__ R, const float* __restrict__ tau,
                                              int N, int g) {
    const int khi = min(N - 3, 32 * T0 + 30), klo = max(0, 32 * T0 - 1);
    for (int kr = khi; kr >= klo; --kr) {
        const float tk = tau[kr];
        if (tk == 0.f) continue;
        const int f = kr + 1;
        const float* rk = R + td_pack_offset(kr, N) - (kr + 2);          // rk[j] = v_j for j >= kr + 2, v_f = 1
        float2 v[MAXT];
        float s;
        {
            const int i0 = 2 * g + 32 * T0, i1 = i0 + 1;
            float a = rk[max(i0, f + 1)], b = rk[max(i1, f + 1)];        // (clamped: never in front of the packed row)
            a = i0 > f ? a : (i0 == f ? 1.f : 0.f);
            b = i1 > f ? b : (i1 == f ? 1.f : 0.f);
            if (32 * T0 + 32 > N) { a = i0 < N ? a : 0.f; b = i1 < N ? b : 0.f; }      // (uniform: MAXT covers a range of N)
            v[T0] = make_float2(a, b);
            s = a * y[T0].x + b * y[T0].y;
        }
#pragma unroll
        for (int t = T0 + 1; t < MAXT; ++t) {
            const int i0 = 2 * g + 32 * t;
            float a = rk[i0], b = rk[i0 + 1];
            if (32 * t + 32 > N) { a = i0 < N ? a : 0.f; b = i0 + 1 < N ? b : 0.f; }
            v[t] = make_float2(a, b);
            s = fmaf(a, y[t].x, s);
            s = fmaf(b, y[t].y, s);
        }
        s = td_row16_sum(s) * tk;
#pragma unroll
        for (int t = T0; t < MAXT; ++t) { y[t].x -= s * v[t].x; y[t].y -= s * v[t].y; }
    }
}
template <int MAXT, int T0>
__device__ __forceinline__ void td_back_all(float2 (&y)[MAXT], const float* __restrict__ R, const float* __restrict__ tau,
                                            int N, int g) {
    td_back_range<MAXT, T0>(y, R, tau, N, g);
    if constexpr (T0 > 0) td_back_all<MAXT, T0 - 1>(y, R, tau, N, g);
}

// MAXT = ceil(N / 32): a vector of length N over the 16 lanes of a DPP row, lane g holds the pairs (2g + 32t, 2g + 32t + 1)
template <int MAXT>
__global__ __launch_bounds__(TD_THREADS) void sym_eig_tridiag_kernel(const float* __restrict__ Lsym, double* __restrict__ bands,
                                                                     float* __restrict__ Q, float* __restrict__ evals,
                                                                     int* __restrict__ sweeps_out, int N, int K, int KP, int ldq,
                                                                     int correct_sign, int areg, long long* __restrict__ prof) {
    extern __shared__ __align__(16) unsigned char td_smem[];
    const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int LD = (N + 3) & ~3;                                 // row pitch of the matrix in LDS (16-byte rows)
    float* A = reinterpret_cast<float*>(td_smem);                // [N][LD]; later packed reflectors + K fp64 vectors (areg floats)
    float* tau = A + areg;                                       // [LD]
    float* es = tau + LD;                                        // [max(2 LD, 256)]: e and 1 / (x0 - beta); later lam, shf
    float* e = es;
    float* scl = es + LD;
    float4* vwx = reinterpret_cast<float4*>(es + max(2 * LD, 256));   // [LD] (v_j, w_j, next x_j, -) of the tridiagonalisation
    float* part = reinterpret_cast<float*>(vwx + LD);            // [1024] matvec partials; (d, e^2); pivot vectors
    double2* de = reinterpret_cast<double2*>(vwx);               // afterwards: [LD] (d_i, e_i)
    double* lam = reinterpret_cast<double*>(es);                 //             [64] eigenvalues (ascending)
    double* shf = lam + 64;                                      //             [64] shifts of the inverse iteration
    const float* Lp = Lsym + (int64_t)p * N * N;
    int stamp = 0;
#define TD_STAMP() do { if (prof && p == 0 && tid == 0) prof[stamp] = (long long)wall_clock64(); ++stamp; } while (0)
    TD_STAMP();

    // ---- A: load (coalesced), symmetrise in LDS; (v, w) = 0 and x = row 0 for the first pass ------------------------------
    for (int idx = tid; idx < N * LD; idx += TD_THREADS) {
        const int i = idx / LD, j = idx - i * LD;
        A[idx] = j < N ? Lp[(int64_t)i * N + j] : 0.f;
    }
    __syncthreads();
    for (int idx = tid; idx < N * LD; idx += TD_THREADS) {       // the thread of (i, j), j > i, owns the pair
        const int i = idx / LD, j = idx - i * LD;
        if (j > i && j < N) {
            const float mij = 0.5f * (A[idx] + A[j * LD + i]);
            A[idx] = mij;
            A[j * LD + i] = mij;
        }
    }
    __syncthreads();
    for (int j = tid; j < LD; j += TD_THREADS) vwx[j] = make_float4(0.f, 0.f, (j >= 1 && j < N) ? A[j] : 0.f, 0.f);
    __syncthreads();
    TD_STAMP();                                                  // 1: loaded

    // ---- B: Householder tridiagonalisation -------------------------------------------------------------------------
    // One pass over the trailing block per reflector: the pass of step k applies reflector k-1 (S -= v w^T + w v^T on rows /
    // columns >= k) and, on the values it has just produced, accumulates t = S x_k over rows >= k+1, where x_k = row k after
    // that update - known beforehand from row k, v and w, so wave 0 prepared it with the previous reflector.  Then wave 0
    // alone: |x|, beta, tau, p = tau S v from t and column k+1 (v = s (x - beta e_1): S v = s (t - beta S e_1)), w, and
    // the next x.  Two barriers and one read + one write of the block per step.
    // Lane = (row phase rs, column quad qd): 16 lanes read 256 contiguous bytes of a row (ds_read_b128, conflict-free),
    // a wave covers 4 rows x 64 columns per iteration; waves = 64-column groups x row chunks.
    const int rs = lane >> 4, qd = lane & 15;
    const int qlast = (N - 1) >> 2;
    long long t_pass = 0, t_p2 = 0, t_bar = 0;
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, vv[4] = {0.f, 0.f, 0.f, 0.f};   // wave 0: x_k and v_k of the step, element f + lane + 64 t
    float r_beta = 0.f, r_tk = 0.f, r_s = 0.f;
    bool r_on = false;
    for (int k = 0; k <= N - 2; ++k) {
        const long long c_a = prof ? (long long)wall_clock64() : 0;
        // wave 0, ahead of its share of the pass: everything of reflector k that only needs x_k (known since the previous
        // step) - |x|, beta, tau, s, v - so that after the barrier only t and column k+1 are missing
        if (wave == 0 && k < N - 2) {
            const int f = k + 1;
            const int tmax = (N - f + 63) >> 6;
            float sig = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xv[t] = 0.f;
                if (t < tmax) {                                  // (uniform)
                    const int ii = lane + 64 * t;
                    const float x = vwx[min(f + ii, N - 1)].z;
                    xv[t] = f + ii < N ? x : 0.f;
                    sig = fmaf(xv[t], ii > 0 ? xv[t] : 0.f, sig);
                }
            }
            sig = cc_wave_sum_fast(sig);
            const float x0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(xv[0])));
            r_beta = x0; r_tk = 0.f; r_s = 0.f; r_on = sig != 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) vv[t] = 0.f;
            if (r_on) {                                          // else nothing to annihilate: H_k = I
                // correctly rounded sqrt / divisions: a one-sided error of an ulp in tau makes every H_k non-orthogonal in the
                // same direction and the eigenvalues drift by 20 ulps of |L| (measured with rcp + Newton); this runs ahead of
                // the pass, off the critical path
                const float nrm = sqrtf(fmaf(x0, x0, sig));
                r_beta = x0 >= 0.f ? -nrm : nrm;
                r_s = 1.f / (x0 - r_beta);                       // same sign as x0: no cancellation
                // tau = (beta - x0) / beta = 2 / |v|^2, |v|^2 = 1 + s^2 sig: H = I - tau v v^T is orthogonal iff tau |v|^2 = 2
                r_tk = 2.f / fmaf(r_s * r_s, sig, 1.f);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (t < tmax) vv[t] = (lane + 64 * t) == 0 ? 1.f : r_s * xv[t];      // (0 beyond N: x = 0 there)
            }
        }
        const int q0 = k >> 2;
        const int nq = qlast - q0 + 1;
        const int cqg = nq > 48 ? 4 : (nq > 32 ? 3 : (nq > 16 ? 2 : 1));
        int cq, ch;                                              // waves = 64-column groups x 4 row chunks (more chunks were
        if (cqg == 4) { cq = wave & 3; ch = wave >> 2; }         // measured slower: the per-wave overhead of a step outweighs
        else if (cqg == 3) { ch = (wave * 11) >> 5; cq = wave - 3 * ch; }   // the shorter row loops, and wave 0 sums more partials)
        else if (cqg == 2) { cq = wave & 1; ch = wave >> 1; }
        else { cq = 0; ch = wave; }
        const int chunks = 4;
        const int rpc = (((N - k + 3) >> 2) + 3) & ~3;           // rows per chunk: ceil(m / 4) rounded to the 4 row phases
        const int stride = 256;
        const int q4 = q0 + (cq << 4) + qd;
        const bool on = ch < chunks && q4 <= qlast;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on) {
            const int jb = k + ch * rpc, je = min(jb + rpc, N);
            const float4 c0 = vwx[4 * q4], c1 = vwx[4 * q4 + 1], c2 = vwx[4 * q4 + 2], c3 = vwx[4 * q4 + 3];
            float* col = A + 4 * q4;
#pragma unroll 2
            for (int j = jb + rs; j < je; j += 4) {
                float4 a = *reinterpret_cast<const float4*>(col + j * LD);
                const float4 o = vwx[j];
                {   // both products rounded before they are added: v_j w_i + w_j v_i is then the same number at (j, i) and
                    // (i, j) and the block stays symmetric to the bit (with an fma it drifts, and t = S^T x != S x)
#pragma clang fp contract(off)
                    a.x -= o.x * c0.y + o.y * c0.x;
                    a.y -= o.x * c1.y + o.y * c1.x;
                    a.z -= o.x * c2.y + o.y * c2.x;
                    a.w -= o.x * c3.y + o.y * c3.x;
                }
                *reinterpret_cast<float4*>(col + j * LD) = a;
                acc.x = fmaf(a.x, o.z, acc.x);
                acc.y = fmaf(a.y, o.z, acc.y);
                acc.z = fmaf(a.z, o.z, acc.z);
                acc.w = fmaf(a.w, o.z, acc.w);
            }
        }
        if (ch < chunks) {                                       // (wave-uniform)
            acc.x = td_sum_rows(acc.x); acc.y = td_sum_rows(acc.y);      // over the 4 row phases (lanes l, l^16, l^32, l^48)
            acc.z = td_sum_rows(acc.z); acc.w = td_sum_rows(acc.w);
            if (on && rs == 0) *reinterpret_cast<float4*>(part + ch * stride + (((cq << 4) + qd) << 2)) = acc;
        }
        const long long c_b = prof ? (long long)wall_clock64() : 0;
        __syncthreads();
        const long long c_c = prof ? (long long)wall_clock64() : 0;
        t_pass += c_b - c_a; t_bar += c_c - c_b;
        if (k == N - 2) break;                                   // the last pass only applies reflector N-3
        if (wave == 0) {
            const int f = k + 1;                                 // first column of reflector k
            const int cbase = q0 << 2;
            const int tmax = (N - f + 63) >> 6;
            float ww[4], cv[4];
            float gam = 0.f;
            const float ts = r_tk * r_s;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ww[t] = 0.f; cv[t] = 0.f;
                if (t < tmax) {                                  // (uniform)
                    const int i = min(f + lane + 64 * t, N - 1), o = i - cbase;
                    const bool in = f + lane + 64 * t < N;
                    const float c = A[f * LD + i];
                    const float tt = (part[o] + part[256 + o]) + (part[512 + o] + part[768 + o]);
                    cv[t] = in ? c : 0.f;
                    ww[t] = in ? ts * (tt - r_beta * c) : 0.f;   // p = tau S v (0 when there is no reflector: ts = 0)
                    gam = fmaf(ww[t], vv[t], gam);
                }
            }
            gam = cc_wave_sum_fast(gam);
            const float hc = 0.5f * r_tk * gam;
#pragma unroll
            for (int t = 0; t < 4; ++t) ww[t] -= hc * vv[t];
            const float w0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ww[0])));
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ii = lane + 64 * t, i = f + ii;
                if (t < tmax && i < N) {                         // .z: row k+1 after this reflector = the next x
                    vwx[i] = make_float4(vv[t], ww[t], ii == 0 ? 0.f : cv[t] - ww[t] - w0 * vv[t], 0.f);
                    // the reflector that is stored must be the x that tau and s were computed from, not the row the pass
                    // produced: the two differ by rounding of terms far larger than a nearly decoupled row (planted
                    // partitions: 1e-4 relative), and H = I - tau v v^T is orthogonal only for the v that tau belongs to
                    A[k * LD + i] = xv[t];
                }
            }
            if (lane == 0) { vwx[k] = make_float4(0.f, 0.f, 0.f, 0.f); e[k] = r_beta; tau[k] = r_tk; scl[k] = r_s; }
        }
        if (prof) t_p2 += (long long)wall_clock64() - c_c;
        __syncthreads();
    }
    // the tridiagonal in fp64 (vwx is no longer read); e_{N-2} is what the last pass left
    {
        double2 mine_de[1];
        const int i = tid;
        if (i < N) {
            const float ei = i < N - 2 ? e[i] : (i == N - 2 ? A[(N - 2) * LD + N - 1] : 0.f);
            mine_de[0] = make_double2((double)A[i * LD + i], (double)ei);
        }
        __syncthreads();                                         // (N <= 196 < 1024: one element per thread)
        if (i < N) de[i] = mine_de[0];
    }
    __syncthreads();
    TD_STAMP();                                                  // 2: tridiagonal
    if (prof && p == 0 && tid == 0) { prof[16] = t_pass; prof[17] = t_bar; prof[18] = t_p2; }

    // ---- pack the reflectors (row k, columns k+2.., scaled) to the front of the region ---------------------------------
    // wave w takes rows w, w + 16, ...; everything is read into registers before anything is written
    const int T = ((N - 1) * (N - 2)) >> 1;
    {
        constexpr int MAXR = 13;                                 // ceil(194 / 16) rows per wave at N = 196
        float reg[MAXR][4];
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int kr = wave + 16 * r;
            const float sc = kr <= N - 3 ? scl[kr] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = kr + 2 + lane + 64 * t;
                reg[r][t] = (kr <= N - 3 && j < N) ? A[kr * LD + j] * sc : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < MAXR; ++r) {
            const int kr = wave + 16 * r;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = kr + 2 + lane + 64 * t;
                if (kr <= N - 3 && j < N) A[td_pack_offset(kr, N) + lane + 64 * t] = reg[r][t];
            }
        }
    }
    const float* R = A;
    double* Y = reinterpret_cast<double*>(A + ((T + 3) & ~3));   // [N][KP]
    double2* ds = reinterpret_cast<double2*>(part);              // [LD] (d_i, max(e_i^2, tiny)) for the Sturm counts
    for (int i = tid; i < N + 24 && i < 256; i += TD_THREADS) {  // (rows beyond N are loaded ahead, never used)
        const double2 t = i < N ? de[i] : make_double2(0.0, 0.0);
        ds[i] = make_double2(t.x, fmax(t.y * t.y, 1e-280));
    }
    __syncthreads();
    TD_STAMP();                                                  // 3: reflectors packed

    // ---- C: eigenvalues (fp64) -----------------------------------------------------------------------------------------
    double glo = 1.0e300, ghi = -1.0e300;
    for (int i = lane; i < N; i += 64) {
        const double r = (i > 0 ? fabs(de[i - 1].y) : 0.0) + fabs(de[i].y);     // (e_{N-1} = 0)
        glo = fmin(glo, de[i].x - r);
        ghi = fmax(ghi, de[i].x + r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { glo = fmin(glo, __shfl_xor(glo, o, 64)); ghi = fmax(ghi, __shfl_xor(ghi, o, 64)); }
    const double tnorm = fmax(fmax(fabs(glo), fabs(ghi)), 1e-300);
    glo -= 4.0 * TD_EPS64 * tnorm * (double)N;
    ghi += 4.0 * TD_EPS64 * tnorm * (double)N;
    if (tid < 4 * K) {                                           // 4 lanes per wanted eigenvalue: one wave per SIMD, the
        const int kb = tid >> 2, gb = tid & 3;                   // recurrence is a dependent chain (latency, not throughput)
        double lo = glo, hi = ghi;
        for (int it = 0; it < TD_SECTIONS; ++it) {
            const double x = lo + (hi - lo) * ((double)(gb + 1) * 0.2);
            const bool below = td_sturm(ds, N, x) <= kb;
            double a = below ? x : lo, b = below ? -hi : -x;
            a = fmax(a, td_dpp_f64<0xB1>(a)); a = fmax(a, td_dpp_f64<0x4E>(a));      // over the quad
            b = fmax(b, td_dpp_f64<0xB1>(b)); b = fmax(b, td_dpp_f64<0x4E>(b));
            lo = a; hi = -b;
        }
        if (gb == 0) lam[kb] = 0.5 * (lo + hi);
    }
    __syncthreads();
    TD_STAMP();                                                  // 4: eigenvalues

    // ---- D: eigenvectors of T (fp64) -------------------------------------------------------------------------------------
    if (tid == 0) {
        // numerically equal eigenvalues get distinct shifts (LAPACK's dstein: 10 ulps of |T|) so that their factorisations
        // differ; the Gram-Schmidt pass makes a basis of the eigenspace out of the solutions
        const double sep = 10.0 * TD_EPS64 * tnorm;
        double prev = lam[0];
        shf[0] = prev;
        for (int q = 1; q < K; ++q) {
            prev = fmax(lam[q], prev + sep);
            shf[q] = prev;
        }
    }
    for (int idx = tid; idx < N * KP; idx += TD_THREADS) {
        const int i = idx / KP, c = idx - i * KP;
        Y[idx] = td_start_value((unsigned)i, (unsigned)c);
    }
    __syncthreads();
    const double tiny = TD_EPS64 * tnorm;
    double* U0 = bands + (int64_t)p * 3 * N * KP;                 // 1 / pivot
    double* U1 = U0 + (int64_t)N * KP;
    double* U2 = U1 + (int64_t)N * KP;
    const int k = tid >> 4, g = tid & 15;                        // 16 lanes per wanted vector
    const bool mine = k < K;
    double2 y[MAXT];
    double* qbuf = reinterpret_cast<double*>(part);              // 2 x [256] pivot vector (ds is no longer read)
    for (int it = 0; it < TD_ITERS; ++it) {
        if (tid < K) {                                           // (T - shift) x = y, lane = eigenvalue
            const int c = tid;
            const double sh = shf[c];
            double2 cur = de[0], nx = de[1];                     // rows i and i + 1 of T; the loads run one row ahead
            double nr = Y[KP + c];
            double a = cur.x - sh, b = cur.y, cc = 0.0, r = Y[c];
            for (int i = 0; i < N - 1; ++i) {
                const int i2 = min(i + 2, N - 1);
                const double2 nx2 = de[i2];
                const double nr2 = Y[i2 * KP + c];
                const double na = cur.y, nb = nx.x - sh, nc = nx.y;                 // (e_{N-1} = 0)
                const bool sw = fabs(na) > fabs(a);
                double pa = sw ? na : a;
                const double pb = sw ? nb : b, pc = sw ? nc : cc, pr = sw ? nr : r;
                const double qa = sw ? a : na, qb = sw ? b : nb, qc = sw ? cc : nc, qr = sw ? r : nr;
                if (fabs(pa) < tiny) pa = pa < 0.0 ? -tiny : tiny;
                double ip = __builtin_amdgcn_rcp(pa);              // v_rcp_f64 + two Newton steps (the IEEE sequence is twice
                ip = fma(fma(-pa, ip, 1.0), ip, ip);                // as long, and this is the dependent chain of the loop)
                ip = fma(fma(-pa, ip, 1.0), ip, ip);
                const double ml = qa * ip;
                U0[i * KP + c] = ip; U1[i * KP + c] = pb; U2[i * KP + c] = pc;
                Y[i * KP + c] = pr;
                a = qb - ml * pb; b = qc - ml * pc; cc = 0.0; r = qr - ml * pr;
                cur = nx; nx = nx2; nr = nr2;
            }
            if (fabs(a) < tiny) a = a < 0.0 ? -tiny : tiny;
            double x1 = r / a, x2 = 0.0;
            Y[(N - 1) * KP + c] = x1;
            // back substitution in blocks of 4 rows, the next block's bands (global) and right-hand sides in flight
            double by[4], b0[4], b1[4], b2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = max(N - 2 - u, 0);
                by[u] = Y[i * KP + c]; b0[u] = U0[i * KP + c]; b1[u] = U1[i * KP + c]; b2[u] = U2[i * KP + c];
            }
            for (int ib = N - 2; ib >= 0; ib -= 4) {
                double cy[4], c0[4], c1[4], c2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { cy[u] = by[u]; c0[u] = b0[u]; c1[u] = b1[u]; c2[u] = b2[u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = max(ib - 4 - u, 0);
                    by[u] = Y[i * KP + c]; b0[u] = U0[i * KP + c]; b1[u] = U1[i * KP + c]; b2[u] = U2[i * KP + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (ib - u >= 0) {
                        const double x = (cy[u] - c1[u] * x1 - c2[u] * x2) * c0[u];
                        Y[(ib - u) * KP + c] = x;
                        x2 = x1; x1 = x;
                    }
                }
            }
        }
        __syncthreads();
        TD_STAMP();                                              // 5, 7: solve
        if (mine) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                const int i0 = 2 * g + 32 * t;
                y[t].x = i0 < N ? Y[i0 * KP + k] : 0.0;
                y[t].y = i0 + 1 < N ? Y[(i0 + 1) * KP + k] : 0.0;
            }
            double mx = 0.0;                                     // keep the numbers small: a solve grows a vector by up to 1 / tiny
#pragma unroll
            for (int t = 0; t < MAXT; ++t) mx = fmax(mx, fmax(fabs(y[t].x), fabs(y[t].y)));
            mx = td_row16_max(mx);
            const double im = mx > 0.0 ? 1.0 / mx : 0.0;
#pragma unroll
            for (int t = 0; t < MAXT; ++t) { y[t].x *= im; y[t].y *= im; }
        }
        for (int kk = 0; kk < K; ++kk) {                         // modified Gram-Schmidt, right-looking
            double* qb = qbuf + (kk & 1) * 256;
            if (mine && k == kk) {
                double n2 = 0.0;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) n2 = fma(y[t].x, y[t].x, fma(y[t].y, y[t].y, n2));
                n2 = td_row16_sum(n2);
                const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int i0 = 2 * g + 32 * t;
                    y[t].x *= inv; y[t].y *= inv;
                    if (i0 < N) *reinterpret_cast<double2*>(qb + i0) = y[t];     // (the pad element of an odd N is 0)
                }
            }
            __syncthreads();
            if (mine && k > kk) {
                double2 qv[MAXT];
                double cf = 0.0;
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int i0 = 2 * g + 32 * t;
                    qv[t] = i0 < N ? *reinterpret_cast<const double2*>(qb + i0) : make_double2(0.0, 0.0);
                    cf = fma(qv[t].x, y[t].x, fma(qv[t].y, y[t].y, cf));
                }
                cf = td_row16_sum(cf);
#pragma unroll
                for (int t = 0; t < MAXT; ++t) { y[t].x -= cf * qv[t].x; y[t].y -= cf * qv[t].y; }
            }
        }
        TD_STAMP();                                              // 6, 8: Gram-Schmidt
        if (it + 1 < TD_ITERS) {
            if (mine) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    const int i0 = 2 * g + 32 * t;
                    if (i0 < N) Y[i0 * KP + k] = y[t].x;
                    if (i0 + 1 < N) Y[(i0 + 1) * KP + k] = y[t].y;
                }
            }
            __syncthreads();
        }
    }

    // ---- E: back-transformation (fp32) ----------------------------------------------------------------------------------
    float2 z[MAXT];
    if (mine) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) z[t] = make_float2((float)y[t].x, (float)y[t].y);
        td_back_all<MAXT, MAXT - 1>(z, R, tau, N, g);
    }
    __syncthreads();                                             // every group is done with Y / the pivot buffers
    TD_STAMP();                                                  // 9: back-transformed

    // ---- F: column order and sign of the reference, coalesced store ------------------------------------------------------
    float* Yf = reinterpret_cast<float*>(Y);                     // [N][KP] fp32 staging
    if (mine) {
        float sg = 0.f;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const float u0 = z[t].x, u1 = z[t].y;
            sg += (u0 > 0.f ? 1.f : (u0 < 0.f ? -1.f : 0.f)) * (u0 * u0) + (u1 > 0.f ? 1.f : (u1 < 0.f ? -1.f : 0.f)) * (u1 * u1);
        }
        sg = td_row16_sum(sg);
        const float flip = correct_sign ? (sg > 0.f ? 1.f : (sg < 0.f ? -1.f : 0.f)) : 1.f;
        const int col = K - 1 - k;
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            const int i0 = 2 * g + 32 * t;
            if (i0 < N) Yf[i0 * KP + col] = z[t].x * flip;
            if (i0 + 1 < N) Yf[(i0 + 1) * KP + col] = z[t].y * flip;
        }
        if (g == 0 && evals) evals[(int64_t)p * K + col] = (float)lam[k];
    }
    __syncthreads();
    float* Qp = Q + (int64_t)p * N * ldq;
    for (int idx = tid; idx < N * K; idx += TD_THREADS) {
        const int i = idx / K, c = idx - i * K;
        Qp[(int64_t)i * ldq + c] = Yf[i * KP + c];
    }
    if (tid == 0 && sweeps_out) sweeps_out[p] = 0;               // a direct method: no sweeps
    TD_STAMP();                                                  // 10: stored
#undef TD_STAMP
}

// ---------------------------------------------------------------------------------------------------------------------
// The same solver for matrices that do not fit in LDS (196 < N <= 832, K <= 192): the matrix lives in a global scratch
// (L2 / MALL resident: 0.6 MB at N = 392), the fused pass streams the trailing block through the CU (float4 rows,
// coalesced), the K vectors are fp64 rows of a second scratch (vector-major: a wave reads a vector contiguously),
// Gram-Schmidt runs from memory (pivot vector in LDS, a wave per remaining vector), the back-transformation keeps the fp32
// vectors in registers (G lanes per vector) and reads the reflector rows where the tridiagonalisation left them.
// Phases, precisions and formulas are those of sym_eig_tridiag_kernel.
__device__ __forceinline__ double td_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#ifndef TDB_ROWS
#define TDB_ROWS 8
#endif
#define TDB_MAXK 192                          // vectors per problem (ViT-B/16 ships K = 160 at N = 784, scripts/activitynet.sh:104-122)
#define TDB_MAXN 832
// MULTI: more vectors than TD_THREADS / G - the back-transformation runs in passes (N > 640 or K > 128)
template <int G, int MAXE, int MAXQ, bool MULTI = false>   // G lanes per vector in the back-transformation, MAXE = ceil(N / G), MAXQ = ceil(N / 64)
__global__ __launch_bounds__(TD_THREADS) void sym_eig_tridiag_big_kernel(const float* __restrict__ Lsym, float* __restrict__ fwork,
                                                                         double* __restrict__ dwork, float* __restrict__ Q,
                                                                         float* __restrict__ evals, int* __restrict__ sweeps_out,
                                                                         int N, int K, int KP, int ldq, int correct_sign,
                                                                         long long fstride, long long dstride, long long* __restrict__ prof) {
    extern __shared__ __align__(16) unsigned char td_smem[];
    const int p = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int LD = (N + 3) & ~3;
    float* A = fwork + (int64_t)p * fstride;                     // [N][LD]
    float* Rf = A + (int64_t)N * ((N + 3) & ~3);                 // [N][LD] reflectors (raw x_k in row k, columns k+1..)
    double* Yv = dwork + (int64_t)p * dstride;                   // [K][LD] vectors
    double* U0 = Yv + (int64_t)K * LD;                           // [N][KP] x 3 bands of U (1 / pivot first)
    double* U1 = U0 + (int64_t)N * KP;
    double* U2 = U1 + (int64_t)N * KP;
    float* tau = reinterpret_cast<float*>(td_smem);              // [LD]
    float* scl = tau + LD;                                       // [LD]
    float* e = scl + LD;                                         // [LD]
    float4* vwx = reinterpret_cast<float4*>(e + LD);             // [LD]; later de
    float* part = reinterpret_cast<float*>(vwx + LD);            // [4][LD + 24]; later ds
    double* lam = reinterpret_cast<double*>(part + 4 * (LD + 24));   // [TDB_MAXK]
    double* shf = lam + TDB_MAXK;                                // [TDB_MAXK]
    double* qbuf = shf + TDB_MAXK;                               // [LD] pivot vector of Gram-Schmidt
    double* red = qbuf + LD;                                     // [16]
    float* cbuf = reinterpret_cast<float*>(red + 16);            // [LD] row k+1 of the step, as the pass produced it
    double2* de = reinterpret_cast<double2*>(vwx);
    const int LDP = LD + 24;
    const float* Lp = Lsym + (int64_t)p * N * N;
    if (prof && p == 0 && tid == 0) prof[0] = (long long)wall_clock64();

    // ---- A: symmetrised copy ------------------------------------------------------------------------------------------
    for (int idx = tid; idx < N * LD; idx += TD_THREADS) {
        const int i = idx / LD, j = idx - i * LD;
        A[idx] = j < N ? 0.5f * (Lp[(int64_t)i * N + j] + Lp[(int64_t)j * N + i]) : 0.f;
    }
    for (int j = tid; j < LD; j += TD_THREADS)
        vwx[j] = make_float4(0.f, 0.f, (j >= 1 && j < N) ? 0.5f * (Lp[j] + Lp[(int64_t)j * N]) : 0.f, 0.f);
    __syncthreads();

    if (prof && p == 0 && tid == 0) prof[1] = (long long)wall_clock64();
    // ---- B: tridiagonalisation (see sym_eig_tridiag_kernel) --------------------------------------------------------------
    // Ownership of the matrix is FIXED: row j belongs to row chunk (j >> 2) & 3 and row phase j & 3, column quad q to column
    // group (q >> 4) & 3 and lane q & 15 - the same lane reads and writes an element in every step, so the global stores of a
    // pass need not be visible to anybody else and the two barriers of a step only order LDS (s_waitcnt lgkmcnt(0) +
    // s_barrier: stores stay in flight).  What other waves need of the fresh block goes through LDS: the partial sums and
    // row k+1, which its owners copy into cbuf as they produce it.
    const int qd = lane & 15, rph = lane >> 4;                   // column quad inside a group, row phase
    const int ch = wave >> 2, cq0 = wave & 3;
    const int qlast = (N - 1) >> 2;
    float xv[MAXQ], vv[MAXQ];
#pragma unroll
    for (int t = 0; t < MAXQ; ++t) { xv[t] = 0.f; vv[t] = 0.f; }
    float r_beta = 0.f, r_tk = 0.f, r_s = 0.f;
    for (int k = 0; k <= N - 2; ++k) {
        if (wave == 0 && k < N - 2) {
            const int f = k + 1;
            float sig = 0.f;
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) {
                const int ii = lane + 64 * t;
                const float x = vwx[min(f + ii, N - 1)].z;
                xv[t] = f + ii < N ? x : 0.f;
                sig = fmaf(xv[t], ii > 0 ? xv[t] : 0.f, sig);
            }
            sig = cc_wave_sum_fast(sig);
            const float x0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(xv[0])));
            r_beta = x0; r_tk = 0.f; r_s = 0.f;
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) vv[t] = 0.f;
            if (sig != 0.f) {
                const float nrm = sqrtf(fmaf(x0, x0, sig));
                r_beta = x0 >= 0.f ? -nrm : nrm;
                r_s = 1.f / (x0 - r_beta);
                r_tk = 2.f / fmaf(r_s * r_s, sig, 1.f);
#pragma unroll
                for (int t = 0; t < MAXQ; ++t) vv[t] = (lane + 64 * t) == 0 ? 1.f : r_s * xv[t];
            }
        }
        const int q0 = k >> 2;
        {
            // my rows: j = 4 ch + rph (mod 16), j >= k
            const int res = 4 * ch + rph;
            const int jfirst = k + ((res - k) & 15);
            for (int ag = (q0 >> 4) + ((cq0 - (q0 >> 4)) & 3); (ag << 4) <= qlast; ag += 4) {
                const int q4 = (ag << 4) + qd;
                const bool on = q4 >= q0 && q4 <= qlast;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (on) {
                    const float4 c0 = vwx[4 * q4], c1 = vwx[4 * q4 + 1], c2 = vwx[4 * q4 + 2], c3 = vwx[4 * q4 + 3];
                    float* col = A + 4 * q4;
                    for (int j8 = jfirst; j8 < N; j8 += 16 * TDB_ROWS) {   // TDB_ROWS rows in flight per lane
                        float4 a[TDB_ROWS];
#pragma unroll
                        for (int u = 0; u < TDB_ROWS; ++u) {
                            const int j = j8 + 16 * u;
                            a[u] = j < N ? *reinterpret_cast<const float4*>(col + (int64_t)j * LD) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
#pragma unroll
                        for (int u = 0; u < TDB_ROWS; ++u) {
                            const int j = j8 + 16 * u;
                            if (j < N) {
                                const float4 o = vwx[j];
                                {
#pragma clang fp contract(off)
                                    a[u].x -= o.x * c0.y + o.y * c0.x;
                                    a[u].y -= o.x * c1.y + o.y * c1.x;
                                    a[u].z -= o.x * c2.y + o.y * c2.x;
                                    a[u].w -= o.x * c3.y + o.y * c3.x;
                                }
                                *reinterpret_cast<float4*>(col + (int64_t)j * LD) = a[u];
                                if (j == k + 1) *reinterpret_cast<float4*>(cbuf + 4 * q4) = a[u];
                                acc.x = fmaf(a[u].x, o.z, acc.x);
                                acc.y = fmaf(a[u].y, o.z, acc.y);
                                acc.z = fmaf(a[u].z, o.z, acc.z);
                                acc.w = fmaf(a[u].w, o.z, acc.w);
                            }
                        }
                    }
                }
                acc.x = td_sum_rows(acc.x); acc.y = td_sum_rows(acc.y);
                acc.z = td_sum_rows(acc.z); acc.w = td_sum_rows(acc.w);
                if (on && rph == 0) *reinterpret_cast<float4*>(part + ch * LDP + 4 * q4) = acc;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (k == N - 2) break;
        if (wave == 0) {
            const int f = k + 1;
            float ww[MAXQ], cv[MAXQ];
            float gam = 0.f;
            const float ts = r_tk * r_s;
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) {
                const int i = min(f + lane + 64 * t, N - 1);
                const bool in = f + lane + 64 * t < N;
                const float c = cbuf[i];
                const float tt = (part[i] + part[LDP + i]) + (part[2 * LDP + i] + part[3 * LDP + i]);
                cv[t] = in ? c : 0.f;
                ww[t] = in ? ts * (tt - r_beta * c) : 0.f;
                gam = fmaf(ww[t], vv[t], gam);
            }
            gam = cc_wave_sum_fast(gam);
            const float hc = 0.5f * r_tk * gam;
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) ww[t] -= hc * vv[t];
            const float w0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(ww[0])));
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) {
                const int ii = lane + 64 * t, i = f + ii;
                if (i < N) {
                    vwx[i] = make_float4(vv[t], ww[t], ii == 0 ? 0.f : cv[t] - ww[t] - w0 * vv[t], 0.f);
                    Rf[(int64_t)k * LD + i] = xv[t];             // the analytic x (the reflector tau belongs to), in its own array:
                                                                 // row k of the matrix has stores of its owners in flight
                }
            }
            if (lane == 0) { vwx[k] = make_float4(0.f, 0.f, 0.f, 0.f); e[k] = r_beta; tau[k] = r_tk; scl[k] = r_s; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();                                             // the matrix is read by other lanes from here on
    {
        double2 mine_de = make_double2(0.0, 0.0);
        if (tid < N) {
            const float ei = tid < N - 2 ? e[tid] : (tid == N - 2 ? A[(int64_t)(N - 2) * LD + N - 1] : 0.f);
            mine_de = make_double2((double)A[(int64_t)tid * LD + tid], (double)ei);
        }
        __syncthreads();
        if (tid < N) de[tid] = mine_de;                          // (N <= 832 < 1024: one element per thread)
    }
    __syncthreads();
    double2* ds = reinterpret_cast<double2*>(part);
    for (int i = tid; i < N + 24; i += TD_THREADS) {
        const double2 t = i < N ? de[i] : make_double2(0.0, 0.0);
        ds[i] = make_double2(t.x, fmax(t.y * t.y, 1e-280));
    }
    __syncthreads();

    if (prof && p == 0 && tid == 0) prof[2] = (long long)wall_clock64();
    // ---- C: eigenvalues ---------------------------------------------------------------------------------------------------
    double glo = 1.0e300, ghi = -1.0e300;
    for (int i = lane; i < N; i += 64) {
        const double r = (i > 0 ? fabs(de[i - 1].y) : 0.0) + fabs(de[i].y);
        glo = fmin(glo, de[i].x - r);
        ghi = fmax(ghi, de[i].x + r);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { glo = fmin(glo, __shfl_xor(glo, o, 64)); ghi = fmax(ghi, __shfl_xor(ghi, o, 64)); }
    const double tnorm = fmax(fmax(fabs(glo), fabs(ghi)), 1e-300);
    glo -= 4.0 * TD_EPS64 * tnorm * (double)N;
    ghi += 4.0 * TD_EPS64 * tnorm * (double)N;
    if (tid < 4 * K) {
        const int kb = tid >> 2, gb = tid & 3;
        double lo = glo, hi = ghi;
        for (int it = 0; it < TD_SECTIONS; ++it) {
            const double x = lo + (hi - lo) * ((double)(gb + 1) * 0.2);
            const bool below = td_sturm(ds, N, x) <= kb;
            double a = below ? x : lo, b = below ? -hi : -x;
            a = fmax(a, td_dpp_f64<0xB1>(a)); a = fmax(a, td_dpp_f64<0x4E>(a));
            b = fmax(b, td_dpp_f64<0xB1>(b)); b = fmax(b, td_dpp_f64<0x4E>(b));
            lo = a; hi = -b;
        }
        if (gb == 0) lam[kb] = 0.5 * (lo + hi);
    }
    __syncthreads();

    if (prof && p == 0 && tid == 0) prof[3] = (long long)wall_clock64();
    // ---- D: eigenvectors of T ------------------------------------------------------------------------------------------------
    if (tid == 0) {
        const double sep = 10.0 * TD_EPS64 * tnorm;
        double prev = lam[0];
        shf[0] = prev;
        for (int q = 1; q < K; ++q) { prev = fmax(lam[q], prev + sep); shf[q] = prev; }
    }
    for (int idx = tid; idx < K * LD; idx += TD_THREADS) {
        const int c = idx / LD, i = idx - c * LD;
        Yv[idx] = i < N ? td_start_value((unsigned)i, (unsigned)c) : 0.0;
    }
    __syncthreads();
    const double tiny = TD_EPS64 * tnorm;
    if (tid < K) {                                               // (T - shift) x = y, lane = eigenvalue
        const int c = tid;
        double* yc = Yv + (int64_t)c * LD;
        const double sh = shf[c];
        double2 cur = de[0], nx = de[1];
        double nr = yc[1];
        double a = cur.x - sh, b = cur.y, cc = 0.0, r = yc[0];
        for (int i = 0; i < N - 1; ++i) {
            const int i2 = min(i + 2, N - 1);
            const double2 nx2 = de[i2];
            const double nr2 = yc[i2];
            const double na = cur.y, nb = nx.x - sh, nc = nx.y;
            const bool sw = fabs(na) > fabs(a);
            double pa = sw ? na : a;
            const double pb = sw ? nb : b, pc = sw ? nc : cc, pr = sw ? nr : r;
            const double qa = sw ? a : na, qb = sw ? b : nb, qc = sw ? cc : nc, qr = sw ? r : nr;
            if (fabs(pa) < tiny) pa = pa < 0.0 ? -tiny : tiny;
            double ip = __builtin_amdgcn_rcp(pa);
            ip = fma(fma(-pa, ip, 1.0), ip, ip);
            ip = fma(fma(-pa, ip, 1.0), ip, ip);
            const double ml = qa * ip;
            U0[i * KP + c] = ip; U1[i * KP + c] = pb; U2[i * KP + c] = pc;
            yc[i] = pr;
            a = qb - ml * pb; b = qc - ml * pc; cc = 0.0; r = qr - ml * pr;
            cur = nx; nx = nx2; nr = nr2;
        }
        if (fabs(a) < tiny) a = a < 0.0 ? -tiny : tiny;
        double x1 = r / a, x2 = 0.0;
        yc[N - 1] = x1;
        for (int i = N - 2; i >= 0; --i) {
            const double x = (yc[i] - U1[i * KP + c] * x1 - U2[i * KP + c] * x2) * U0[i * KP + c];
            yc[i] = x;
            x2 = x1; x1 = x;
        }
    }
    __syncthreads();
    if (prof && p == 0 && tid == 0) prof[4] = (long long)wall_clock64();
    for (int kk = 0; kk < K; ++kk) {                             // modified Gram-Schmidt from memory
        double* yk = Yv + (int64_t)kk * LD;
        const double v = tid < N ? yk[tid] : 0.0;
        const double s = td_wave_sum(v * v);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        double n2 = 0.0;
#pragma unroll
        for (int w = 0; w < TD_WAVES; ++w) n2 += red[w];
        const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
        if (tid < N) { const double qv = v * inv; qbuf[tid] = qv; yk[tid] = qv; }
        __syncthreads();
        for (int k2 = kk + 1 + wave; k2 < K; k2 += TD_WAVES) {    // a wave per remaining vector
            double* y2 = Yv + (int64_t)k2 * LD;
            double yv[MAXQ];
            double dsum = 0.0;
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) {
                const int i = lane + 64 * t;
                yv[t] = i < N ? y2[i] : 0.0;
                dsum = fma(i < N ? qbuf[i] : 0.0, yv[t], dsum);
            }
            dsum = td_wave_sum(dsum);
#pragma unroll
            for (int t = 0; t < MAXQ; ++t) {
                const int i = lane + 64 * t;
                if (i < N) y2[i] = yv[t] - dsum * qbuf[i];
            }
        }
        __syncthreads();
    }

    if (prof && p == 0 && tid == 0) prof[5] = (long long)wall_clock64();
    // ---- E: back-transformation, fp32 vectors in registers (G lanes per vector), TD_THREADS / G vectors per pass over the
    // reflectors (K = 160 at N = 784: three passes of 64 vectors - the vectors of a pass stay in registers for its whole sweep)
    auto back_pass = [&](int kpass) {
    const int k = kpass + tid / G, g = tid % G;
    const bool mine = k < K;
    float z[MAXE];
    if (mine) {
#pragma unroll
        for (int t = 0; t < MAXE; ++t) { const int i = g + G * t; z[t] = i < N ? (float)Yv[(int64_t)k * LD + i] : 0.f; }
    }
    {
        // every reflector goes through LDS once (0 ... 0, 1, v ...), the next one is fetched while this one is applied
        float* vb = part;                                        // 2 x [LD]
        auto fetch = [&](int kr) -> float {
            if (kr < 0 || tid >= LD) return 0.f;
            const int f = kr + 1;
            const float raw = (tid > f && tid < N) ? Rf[(int64_t)kr * LD + tid] * scl[kr] : 0.f;
            return tid == f ? 1.f : raw;
        };
        float r1 = fetch(N - 3);
        if (tid < LD) vb[((N - 3) & 1) * LD + tid] = r1;
        r1 = fetch(N - 4);                                       // two rows ahead: the loads stay in flight across the barriers
        __syncthreads();
        for (int kr = N - 3; kr >= 0; --kr) {
            const float r2 = fetch(kr - 2);
            const float tk = tau[kr];
            if (mine && tk != 0.f) {
                const int f = kr + 1;
                const float* v = vb + (kr & 1) * LD;
                float s = 0.f;
#pragma unroll
                for (int t = 0; t < MAXE; ++t) {
                    if (G * t + G - 1 < f) continue;             // (uniform: the reflector is zero there)
                    const int i = g + G * t;
                    s = fmaf(i < LD ? v[i] : 0.f, z[t], s);
                }
                s += cc_dpp_f32<0xB1>(s); s += cc_dpp_f32<0x4E>(s); s += cc_dpp_f32<0x141>(s);
                if (G == 16) s += cc_dpp_f32<0x140>(s);
                s *= tk;
#pragma unroll
                for (int t = 0; t < MAXE; ++t) {
                    if (G * t + G - 1 < f) continue;
                    const int i = g + G * t;
                    z[t] -= s * (i < LD ? v[i] : 0.f);
                }
            }
            if (kr > 0 && tid < LD) vb[((kr - 1) & 1) * LD + tid] = r1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            r1 = r2;
        }
    }
    if (mine) {
        // ---- F
        float sg = 0.f;
#pragma unroll
        for (int t = 0; t < MAXE; ++t) { const float u = z[t]; sg += (u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f)) * (u * u); }
        sg += cc_dpp_f32<0xB1>(sg); sg += cc_dpp_f32<0x4E>(sg); sg += cc_dpp_f32<0x141>(sg);
        if (G == 16) sg += cc_dpp_f32<0x140>(sg);
        const float flip = correct_sign ? (sg > 0.f ? 1.f : (sg < 0.f ? -1.f : 0.f)) : 1.f;
        const int col = K - 1 - k;
        float* Qp = Q + (int64_t)p * N * ldq;
#pragma unroll
        for (int t = 0; t < MAXE; ++t) { const int i = g + G * t; if (i < N) Qp[(int64_t)i * ldq + col] = z[t] * flip; }
        if (g == 0 && evals) evals[(int64_t)p * K + col] = (float)lam[k];
    }
    };
    if constexpr (MULTI) {
        for (int kpass = 0; kpass < K; kpass += TD_THREADS / G) {
            back_pass(kpass);
            __syncthreads();                                     // the reflector strip is refilled by the next pass
        }
    } else {
        back_pass(0);
    }
    if (tid == 0 && sweeps_out) sweeps_out[p] = 0;
    __syncthreads();
    if (prof && p == 0 && tid == 0) prof[6] = (long long)wall_clock64();
}

// floats of the big region: the matrix, later the packed reflectors (fp32) + K vectors (fp64)
size_t td_region_floats(int N, int K) {
    const size_t LD = (size_t)((N + 3) & ~3), KP = (size_t)(K | 1);
    const size_t T = ((size_t)(N - 1) * (N - 2)) >> 1;
    const size_t after = ((T + 3) & ~(size_t)3) + 2 * (size_t)N * KP;
    return (((size_t)N * LD > after ? (size_t)N * LD : after) + 3) & ~(size_t)3;
}
size_t td_smem_bytes(int N, int K) {
    const size_t LD = (size_t)((N + 3) & ~3);
    const size_t es = 2 * LD > 256 ? 2 * LD : 256;
    return (td_region_floats(N, K) + LD + es + 4 * LD + 1024) * sizeof(float);
}

long long* g_td_prof = nullptr;

}  // namespace

// development builds (-DCC_DEV_KNOBS) only: wall-clock stamps (100 MHz) of workgroup 0 at the phase boundaries.
#ifdef CC_DEV_KNOBS
extern "C" void cc_debug_set_eig_profile(long long* dev_buf) { g_td_prof = dev_buf; }
#endif

bool cc_sym_eig_tridiag_supports(int N, int K) {
    if (N < 3 || N > 196 || K < 1 || K > 64 || K > N) return false;
    return td_smem_bytes(N, K) <= 160 * 1024;                    // (N = 196: K <= 49)
}

size_t cc_sym_eig_tridiag_ws_bytes(int P, int N) {               // three fp64 bands of U for K <= 64 lanes per problem
    return cc_align_up((size_t)P * 3 * N * 65 * sizeof(double), 256);
}

// ---- large N: per-problem scratch = matrix [N][LD] floats, then doubles: K vectors [LD] + 3 bands [N][KP]
static size_t tdb_fstride(int N) { return 2 * (size_t)N * ((N + 3) & ~3); }      // matrix + reflectors
static size_t tdb_dstride(int N, int K) { return (size_t)K * ((N + 3) & ~3) + 3 * (size_t)N * (K | 1); }
static size_t tdb_smem_bytes(int N) {
    const size_t LD = (size_t)((N + 3) & ~3);
    return (3 * LD + 4 * LD + 4 * (LD + 24) + LD) * sizeof(float) + (2 * TDB_MAXK + LD + 16) * sizeof(double);
}
bool cc_sym_eig_tridiag_big_supports(int N, int K) { return N > 196 && N <= TDB_MAXN && K >= 1 && K <= TDB_MAXK && K <= N; }

size_t cc_sym_eig_tridiag_big_ws_bytes(int P, int N) {           // (sized for K = TDB_MAXK)
    return cc_align_up((size_t)P * tdb_fstride(N) * sizeof(float), 256) +
           cc_align_up((size_t)P * tdb_dstride(N, TDB_MAXK) * sizeof(double), 256);
}

int cc_launch_sym_eig_tridiag(const float* laplacian, int P, int N, int K, int correct_sign, float* Q, int ldq, float* evals,
                              int* sweeps_out, void* ws, size_t ws_bytes, hipStream_t st) {
    if (cc_sym_eig_tridiag_big_supports(N, K)) {
        if (!ws || ws_bytes < cc_sym_eig_tridiag_big_ws_bytes(P, N)) return CC_ERR_WORKSPACE;
        float* fw = static_cast<float*>(ws);
        double* dw = reinterpret_cast<double*>(static_cast<unsigned char*>(ws) + cc_align_up((size_t)P * tdb_fstride(N) * sizeof(float), 256));
        const size_t smem = tdb_smem_bytes(N);
        const int KP = K | 1;
#define TDB_LAUNCH(G, MAXE, MAXQ, MULTI)                                                                                \
    do {                                                                                                               \
        auto kern = sym_eig_tridiag_big_kernel<G, MAXE, MAXQ, MULTI>;                                                   \
        hipLaunchKernelGGL(kern, dim3(P), dim3(TD_THREADS), smem, st, laplacian, fw, dw, Q, evals, sweeps_out, N, K, KP, ldq, \
                           correct_sign, (long long)tdb_fstride(N), (long long)tdb_dstride(N, K), g_td_prof);                     \
    } while (0)
        if (N > 640 || K > 128) {                   // passes of 64 vectors (98 registers per vector at G = 8, N = 784 would not fit)
            if (N <= 320) TDB_LAUNCH(16, 20, 5, true);
            else if (N <= 448) TDB_LAUNCH(16, 28, 7, true);
            else if (N <= 640) TDB_LAUNCH(16, 40, 10, true);
            else TDB_LAUNCH(16, 52, 13, true);
        } else if (K <= 64) {
            if (N <= 320) TDB_LAUNCH(16, 20, 5, false);
            else if (N <= 448) TDB_LAUNCH(16, 28, 7, false);
            else TDB_LAUNCH(16, 40, 10, false);
        } else {
            if (N <= 320) TDB_LAUNCH(8, 40, 5, false);
            else if (N <= 448) TDB_LAUNCH(8, 56, 7, false);
            else TDB_LAUNCH(8, 80, 10, false);
        }
#undef TDB_LAUNCH
        CC_LAUNCH_CHECK();
        return CC_OK;
    }
    if (!cc_sym_eig_tridiag_supports(N, K)) return CC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < cc_sym_eig_tridiag_ws_bytes(P, N)) return CC_ERR_WORKSPACE;
    const int KP = K | 1;
    const size_t smem = td_smem_bytes(N, K);
    const int areg = (int)td_region_floats(N, K);
#define TD_LAUNCH(MAXT)                                                                                                 \
    do {                                                                                                               \
        auto kern = sym_eig_tridiag_kernel<MAXT>;                                                                      \
        if (smem > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) \
            return CC_ERR_HIP;                                                                                         \
        hipLaunchKernelGGL(kern, dim3(P), dim3(TD_THREADS), smem, st, laplacian, static_cast<double*>(ws), Q, evals,    \
                           sweeps_out, N, K, KP, ldq, correct_sign, areg, g_td_prof);                                        \
    } while (0)
    if (N <= 64) TD_LAUNCH(2);
    else if (N <= 128) TD_LAUNCH(4);
    else TD_LAUNCH(7);
#undef TD_LAUNCH
    CC_LAUNCH_CHECK();
    return CC_OK;
}
