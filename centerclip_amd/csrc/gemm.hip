// fp16-operand / fp32-accumulate GEMM for the CLIP transformer blocks on gfx950.
//
//   C[M,N] (+)= epilogue( A[M,K] (fp16, row-major)  x  W[N,K]^T (fp16, nn.Linear layout) + bias[N] )
//
// Both operands are K-contiguous, which is exactly the MFMA fragment shape
// (v_mfma_f32_16x16x32_f16: 8 consecutive k per lane), so neither is transposed.
// Structure: 256 threads = 4 waves (2x2), BK = 64, tiles staged HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR round trip), LDS image XOR-swizzled through the per-lane
// SOURCE address (the LDS-DMA destination is lane-linear), double-buffered.
// The MFMA is issued with the weight fragment as the A operand and the activation fragment as
// the B operand, so an accumulator lane holds 4 consecutive output features of one row:
// epilogue stores are 8-byte (fp16) / 16-byte (fp32) instead of 2-byte scatters.
//
// Epilogues (fused): bias -> fp16 (QKV), bias + QuickGELU -> fp16 (c_fc),
// bias + residual add -> fp32 in place (out_proj / c_proj), + positional embedding with the
// patch-row -> token-row remap (conv1 as im2col GEMM, modules/clip.py:282,324-336).
#include "cc_kernels.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BK 64

__device__ __forceinline__ void glds16(const _Float16* g, _Float16* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ float quick_gelu(float x) {      // modules/clip.py:192-194
    return x / (1.0f + __expf(-1.702f * x));
}

template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs g) {
    constexpr int MI = BM / 32, NI = BN / 32;            // 16x16 fragments per wave (wave tile BM/2 x BN/2)
    constexpr int A_BYTES = BM * GEMM_BK * 2, B_BYTES = BN * GEMM_BK * 2;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_BYTES + B_BYTES)];

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of
    // tiles (tn fastest) so tiles sharing an A panel hit the same L2.
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / g.tiles_n, tn = bid - tm * g.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging addresses: LDS chunk idx -> (row r, chunk position cp); source chunk = cp ^ (r & 7)
    constexpr int A_LOADS = BM * 8 / 256, B_LOADS = BN * 8 / 256;
    const _Float16* asrc[A_LOADS];
    const _Float16* bsrc[B_LOADS];
#pragma unroll
    for (int q = 0; q < A_LOADS; ++q) {
        const int idx = (q * 4 + wave) * 64 + lane, r = idx >> 3, c = (idx & 7) ^ (r & 7);
        asrc[q] = g.A + (int64_t)min(row0 + r, g.M - 1) * g.K + c * 8;
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
        const int idx = (q * 4 + wave) * 64 + lane, r = idx >> 3, c = (idx & 7) ^ (r & 7);
        bsrc[q] = g.W + (int64_t)(col0 + r) * g.K + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        _Float16* la = reinterpret_cast<_Float16*>(smem + buf * (A_BYTES + B_BYTES));
        _Float16* lb = reinterpret_cast<_Float16*>(smem + buf * (A_BYTES + B_BYTES) + A_BYTES);
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q) glds16(asrc[q] + kt * GEMM_BK, la + (q * 4 + wave) * 512);
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) glds16(bsrc[q] + kt * GEMM_BK, lb + (q * 4 + wave) * 512);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / GEMM_BK;
    stage(0, 0);
    __syncthreads();
    const int l15 = lane & 15, lg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) stage(buf ^ 1, kt + 1);
        const unsigned char* la = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            h8 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wr * (BM / 2) + i * 16 + l15;
                af[i] = *reinterpret_cast<const h8*>(la + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = wc * (BN / 2) + j * 16 + l15;
                bf[j] = *reinterpret_cast<const h8*>(lb + r * 128 + (((ks * 4 + lg) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: lane holds C[m = .. + l15][n = .. + lg*4 + 0..3]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = row0 + wr * (BM / 2) + i * 16 + l15;
        if (m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = col0 + wc * (BN / 2) + j * 16 + lg * 4;
            f32x4 v = acc[i][j];
            if (g.bias) {
                const float4 b = *reinterpret_cast<const float4*>(g.bias + n);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (EPI == EPI_F16 || EPI == EPI_F16_GELU) {
                if (EPI == EPI_F16_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                }
                h4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(g.C) + (int64_t)m * g.ldc + n) = o;
            } else if (EPI == EPI_F32_RESID) {
                float* dst = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
                float4 c = *reinterpret_cast<const float4*>(dst);
                c.x += v[0]; c.y += v[1]; c.z += v[2]; c.w += v[3];
                *reinterpret_cast<float4*>(dst) = c;
            } else if (EPI == EPI_F32_PATCH) {
                const int f = m / g.patch_n, tok = m - f * g.patch_n + 1;
                const float4 pe = *reinterpret_cast<const float4*>(g.pos + (int64_t)tok * g.N + n);
                float* dst = reinterpret_cast<float*>(g.C) + ((int64_t)f * (g.patch_n + 1) + tok) * g.ldc + n;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0] + pe.x, v[1] + pe.y, v[2] + pe.z, v[3] + pe.w);
            } else {
                float* dst = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + n;
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

namespace {

template <int BM, int BN>
int launch_tile(GemmArgs g, int epi, hipStream_t st) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = g.N / BN;
    dim3 grid(g.tiles_m * g.tiles_n), block(256);
    switch (epi) {
        case EPI_F16: hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, EPI_F16>), grid, block, 0, st, g); break;
        case EPI_F16_GELU: hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, EPI_F16_GELU>), grid, block, 0, st, g); break;
        case EPI_F32_RESID: hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, EPI_F32_RESID>), grid, block, 0, st, g); break;
        case EPI_F32_PATCH: hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, EPI_F32_PATCH>), grid, block, 0, st, g); break;
        case EPI_F32: hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, EPI_F32>), grid, block, 0, st, g); break;
        default: return CC_ERR_INVALID;
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
}

}  // namespace

// tile: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64
int cc_gemm_dispatch(GemmArgs g, int epi, int tile, hipStream_t st) {
    if (g.M <= 0 || g.N <= 0 || g.K <= 0 || (g.K % GEMM_BK) || (g.N % 64)) return CC_ERR_INVALID;
    if (tile == 0) {
        // fill >= ~2 tiles per CU when possible; otherwise shrink the tile for parallelism
        const long t128 = (long)((g.M + 127) / 128) * (g.N / 128);
        const long t12864 = (long)((g.M + 127) / 128) * (g.N / 64);
        if ((g.N % 128) == 0 && t128 >= 512) tile = 1;
        else if (t12864 >= 512) tile = 2;
        else tile = 4;
    }
    if ((tile == 1 || tile == 3) && (g.N % 128)) return CC_ERR_INVALID;
    switch (tile) {
        case 1: return launch_tile<128, 128>(g, epi, st);
        case 2: return launch_tile<128, 64>(g, epi, st);
        case 3: return launch_tile<64, 128>(g, epi, st);
        case 4: return launch_tile<64, 64>(g, epi, st);
        default: return CC_ERR_INVALID;
    }
}

extern "C" {

int cc_linear_f16(const void* a_f16, const void* w_f16, const float* bias, void* c, int32_t M, int32_t N, int32_t K,
                  int32_t ldc, int32_t epilogue, int32_t tile, void* stream) {
    if (!a_f16 || !w_f16 || !c) return CC_ERR_INVALID;
    if (epilogue == EPI_F32_PATCH) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(a_f16);
    g.W = static_cast<const _Float16*>(w_f16);
    g.bias = bias;
    g.C = c;
    g.M = M; g.N = N; g.K = K; g.ldc = ldc;
    return cc_gemm_dispatch(g, epilogue, tile, static_cast<hipStream_t>(stream));
}

}  // extern "C"
