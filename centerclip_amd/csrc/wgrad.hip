// Weight gradient of a Linear layer without transposed operand copies (N4, round 5).
//
//   dW[n1][n2] = (1 / scale) * sum_m dY[m][n1] * X[m][n2]          (main.py:321 backward of y = x W^T, per layer)
//
// dY [M, N1] and X [M, N2] are the row-major fp16 matrices the backward already holds (the scaled gradient, the saved
// activation): the contraction runs over their ROW index.  Rounds 3-4 fed this product to the forward GEMM kernel (both operands
// contraction-contiguous), which needed dY^T and X^T written out first - 148 transposing launches per step, 1.3 ms.  Here the
// operand tiles are staged as they lie in memory, [32 rows m][128 columns] by LDS-DMA, and the MFMA fragments - a lane needs
// eight consecutive m for one column - come out of the LDS transposing read of gfx950, ds_read_b64_tr_b16: the 16 lanes of a
// group hand in the addresses of sixteen 8-byte pieces (4 rows x 16 columns) and receive the 4 x 16 block column by column.
//
// LDS image of an operand tile: [32 m][16 chunks of 16 bytes], chunk c of row m at position c ^ swz(m) with
// swz(m) = 2 * ((m & 3) | ((m >> 3) & 1) << 2): the eight rows a half-wave reads in one cycle (m0 .. m0 + 3 and m0 + 8 .. m0 + 11)
// then land their 32-byte segments in eight different quarters of the 64 banks.  The LDS-DMA writes 1 KB per wave instruction
// linearly (lane i -> byte 16 i), so the swizzle is applied on the source side: lane i fetches chunk (i & 15) ^ swz(row).
//
// Work split: 128 x 128 output tiles x S slices of the M rows (S chosen so that the grid fills the chip about twice).  S = 1: the
// tile is scaled and stored; S > 1: every slice writes its partial tile and a second kernel adds the S partials of an element
// in slice order - a fixed order, bit-stable from run to run - and scales.
// Loop: four stage buffers of 32 rows (16 KB each: two 64 KB workgroups per CU), three stages in flight; a step waits for the next
// stage only (counted s_waitcnt - the LDS-DMA loads of the two younger stages stay in flight) and meets the other waves at a raw
// s_barrier.  No other vector-memory operation is issued inside the loop, so the count is exact.
#include "cc_common.h"
#include "cc_kernels.h"
#include <type_traits>

namespace {

typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 wg_h8 __attribute__((ext_vector_type(8)));
typedef short wg_s4 __attribute__((ext_vector_type(4)));
typedef _Float16 wg_h4 __attribute__((ext_vector_type(4)));

constexpr int WG_BN = 128;        // output tile: 128 rows of dW (columns of dY) x 128 columns (columns of X)
constexpr int WG_BK = 32;         // rows of dY / X per stage
constexpr int WG_NST = 4;         // stage buffers
constexpr int WG_TILE_BYTES = WG_BK * WG_BN * 2;   // 8 KB per operand and stage
constexpr int WG_LOADS = 4;       // LDS-DMA instructions per wave and stage (2 per operand)

__device__ __forceinline__ void wg_glds16(const _Float16* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int wg_swz(int m) { return 2 * ((m & 3) | (((m >> 3) & 1) << 2)); }

struct WgradArgs {
    const _Float16* dy; const _Float16* x;      // [M][N1], [M][N2]
    float* dw;                                  // [N1][N2]
    float* partial;                             // [S][N1][N2]
    const float* scale;                         // device scalar: the power of two dY was multiplied by (NULL: 1)
    int M, N1, N2, S, rows_per_slice;           // rows_per_slice: a multiple of WG_BK
};

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [WG_NST stages][dY tile | X tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tn2 = a.N2 / WG_BN;
    const int tile = (int)blockIdx.x / a.S, slice = (int)blockIdx.x % a.S;
    const int t1 = tile / tn2, t2 = tile % tn2;
    const int m_begin = slice * a.rows_per_slice, m_end = min(a.M, m_begin + a.rows_per_slice);
    const int nk = __builtin_amdgcn_readfirstlane((m_end - m_begin + WG_BK - 1) / WG_BK);

    // ---- staging: wave instruction q of a wave covers rows 4 (q * 4 + wave) .. + 3 of the stage, lane i -> row + i / 16,
    // position i % 16; 2 instructions per operand and wave.  Rows behind the slice's end (its last stage only) re-read the
    // last row - always one instruction per piece, so the counted waits below stay exact; their fragment entries are cleared
    // in registers before the MFMAs of that step.
    const int srow = lane >> 4, spos = lane & 15;
    int64_t dyoff[2], xoff[2];                           // element offsets of this lane's pieces at row 0
    int srel[2];                                         // row of the stage
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        srel[q] = (q * 4 + wave) * 4 + srow;
        const int c = spos ^ wg_swz(srel[q]);
        dyoff[q] = (int64_t)t1 * WG_BN + c * 8;
        xoff[q] = (int64_t)t2 * WG_BN + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* ldy = smem + buf * 2 * WG_TILE_BYTES;
        unsigned char* lx = ldy + WG_TILE_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = min(m_begin + kt * WG_BK + srel[q], m_end - 1);
            wg_glds16(a.dy + (int64_t)m * a.N1 + dyoff[q], ldy + (q * 4 + wave) * 1024);
            wg_glds16(a.x + (int64_t)m * a.N2 + xoff[q], lx + (q * 4 + wave) * 1024);
        }
    };
    // ---- fragments: wave (wr, wc) owns dW rows 64 wr .. + 63 (dY columns) x columns 64 wc .. + 63 (X columns)
    const int wr = wave >> 1, wc = wave & 1;
    const int g = lane >> 4, i16 = lane & 15;
    // piece of this lane inside a 4 x 16 block: row i16 / 4, 8-byte column (i16 & 3); the two reads of a fragment (rows 8 g + 0 .. 3
    // and 8 g + 4 .. 7) differ in the row and in the swizzle.  The byte address of every read inside a stage buffer is fixed per
    // lane (16 registers); the buffer and the operand come in as the instruction's immediate offset (the loop is unrolled over
    // the four buffers).
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    unsigned ay[4][2], ax[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = g * 8 + h * 4 + (i16 >> 2);
        const unsigned base = lds0 + m * (WG_BN * 2) + ((i16 & 3) & 1) * 8;
        const int sw = wg_swz(m), ch = (i16 & 3) >> 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ay[i][h] = base + ((((wr * 4 + i) * 2 + ch) ^ sw) << 4);
            ax[i][h] = base + ((((wc * 4 + i) * 2 + ch) ^ sw) << 4);
        }
    }
    // The transposing reads are issued as inline assembly: the compiler orders an LDS read it knows of behind EVERY LDS-DMA load in
    // flight (s_waitcnt vmcnt(0) in front of the first fragment read: the three-stage pipeline would collapse to one) - the
    // hand-over of a landed stage is the counted wait + barrier below.  The waits for the reads themselves are asm statements
    // that take the fragments as read-write operands, so that no MFMA can be scheduled in front of them.
    auto join = [](const wg_s4 (&part)[2]) -> wg_h8 {
        return __builtin_shufflevector(__builtin_bit_cast(wg_h4, part[0]), __builtin_bit_cast(wg_h4, part[1]), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    wg_f32x4 acc[4][4];                                  // [dY column block][X column block]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    // valid rows of the slice's last stage (32 = all): entries of a fragment are rows 8 g + 0 .. 7
    const int tail_rows = (m_end - m_begin) - (nk - 1) * WG_BK;
    auto step = [&](auto buf_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr int OY = BUF * 2 * WG_TILE_BYTES, OX = OY + WG_TILE_BYTES;
        if (kt + 3 < nk) stage((BUF + 3) & (WG_NST - 1), kt + 3);      // into the buffer step kt - 1 read (behind its barrier)
        // fragment reads in the order the MFMAs want them (dY block 0, the four X blocks, then dY blocks 1 .. 3); the MFMAs of dY
        // block i start when its two reads have returned (in-order LDS returns: lgkmcnt(6 - 2 i)), under the reads still in flight
        wg_s4 py[4][2], px[4][2];
#define WG_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
        WG_TR(py[0][0], ay[0][0], OY); WG_TR(py[0][1], ay[0][1], OY);
#pragma unroll
        for (int j = 0; j < 4; ++j) { WG_TR(px[j][0], ax[j][0], OX); WG_TR(px[j][1], ax[j][1], OX); }
#pragma unroll
        for (int i = 1; i < 4; ++i) { WG_TR(py[i][0], ay[i][0], OY); WG_TR(py[i][1], ay[i][1], OY); }
#undef WG_TR
        asm volatile("s_waitcnt lgkmcnt(6)"
                     : "+v"(py[0][0]), "+v"(py[0][1]), "+v"(px[0][0]), "+v"(px[0][1]), "+v"(px[1][0]), "+v"(px[1][1]), "+v"(px[2][0]),
                       "+v"(px[2][1]), "+v"(px[3][0]), "+v"(px[3][1])
                     :
                     : "memory");
        const bool tail = (kt == nk - 1) && tail_rows < WG_BK;          // wave-uniform
        auto clear_tail = [&](wg_s4 (&part)[2]) {                        // rows 8 g + 4 h + e >= tail_rows -> 0
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (g * 8 + h * 4 + e >= tail_rows) part[h][e] = 0;
        };
        if (tail) {
#pragma unroll
            for (int j = 0; j < 4; ++j) clear_tail(px[j]);
        }
        wg_h8 fx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) fx[j] = join(px[j]);
        auto mma_row = [&](int i) {
            // A operand = the X fragment (its 16 columns are the rows 4 g + r of the result a lane holds), B = dY:
            // acc[i][j][r] = dW[row: dY column 16 i + i16][column: X column 16 j + 4 g + r]
            const wg_h8 fy = join(py[i]);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fx[j], fy, acc[i][j], 0, 0, 0);
        };
        mma_row(0);
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(py[1][0]), "+v"(py[1][1]) : : "memory");
        mma_row(1);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(py[2][0]), "+v"(py[2][1]) : : "memory");
        mma_row(2);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(py[3][0]), "+v"(py[3][1]) : : "memory");
        mma_row(3);
        // stage kt + 1 landed (the younger ones stay in flight), every wave done reading buffer kt
        if (kt + 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WG_LOADS) : "memory");
        else if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };

    // prologue: stages 0 .. 2 requested, stage 0 landed
    if (nk > 0) stage(0, 0);
    if (nk > 1) stage(1, 1);
    if (nk > 2) stage(2, 2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WG_LOADS) : "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WG_LOADS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int kt0 = 0; kt0 < nk; kt0 += WG_NST) {
        step(std::integral_constant<int, 0>{}, kt0);
        if (kt0 + 1 < nk) step(std::integral_constant<int, 1>{}, kt0 + 1);
        if (kt0 + 2 < nk) step(std::integral_constant<int, 2>{}, kt0 + 2);
        if (kt0 + 3 < nk) step(std::integral_constant<int, 3>{}, kt0 + 3);
    }

    const float inv = (a.S == 1 && a.scale) ? 1.0f / *a.scale : 1.0f;
    float* out = a.S == 1 ? a.dw : a.partial + (int64_t)slice * a.N1 * a.N2;
    const int row0 = t1 * WG_BN + wr * 64, col0 = t2 * WG_BN + wc * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 v = make_float4(acc[i][j][0] * inv, acc[i][j][1] * inv, acc[i][j][2] * inv, acc[i][j][3] * inv);
            *reinterpret_cast<float4*>(out + (int64_t)(row0 + i * 16 + i16) * a.N2 + col0 + j * 16 + g * 4) = v;
        }
}

// dW = (sum over the slices, in slice order) / scale: a float4 per thread, all S loads of a thread in flight.  The workgroups
// behind the first main_blocks add the bias gradient's partial column sums (cc_cast_transpose_f16 left one row of sums per 64
// rows of dY) exactly as the launch this folds away did (column_reduce_kernel: eight segments of the chunks, each added in chunk
// order, then the segment sums in segment order).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int64_t n4, int S,
                                                           const float* __restrict__ scale, int main_blocks,
                                                           const float* __restrict__ col_partial, int chunks, int cols,
                                                           float* __restrict__ db) {
    if ((int)blockIdx.x >= main_blocks) {                        // 32 columns x 8 chunk segments per workgroup
        __shared__ float red[8][32];
        const int cl = threadIdx.x & 31, seg = threadIdx.x >> 5;
        const int c = ((int)blockIdx.x - main_blocks) * 32 + cl;
        const int per = (chunks + 7) / 8, i0 = seg * per, i1 = min(chunks, i0 + per);
        float s = 0.f;
        if (c < cols)
            for (int i = i0; i < i1; ++i) s += col_partial[(int64_t)i * cols + c];
        red[seg][cl] = s;
        __syncthreads();
        if (seg == 0 && c < cols) {
            float t = red[0][cl];
#pragma unroll
            for (int k = 1; k < 8; ++k) t += red[k][cl];
            db[c] = t;
        }
        return;
    }
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n4) return;
    const float4* src = reinterpret_cast<const float4*>(partial) + e;
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < S; s0 += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (s0 + u < S) ? src[(int64_t)(s0 + u) * n4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s0 + u < S) { sum.x += v[u].x; sum.y += v[u].y; sum.z += v[u].z; sum.w += v[u].w; }
    }
    const float inv = scale ? 1.0f / *scale : 1.0f;
    reinterpret_cast<float4*>(dw)[e] = make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
}

int wgrad_slices(int M, int N1, int N2) {
    const long tiles = (long)(N1 / WG_BN) * (N2 / WG_BN);
    long s = (2 * 256) / tiles;                                          // at most two workgroups per CU: one round
    const long max_s = (M + 8 * WG_BK - 1) / (8 * WG_BK);                // at least eight stages per slice
    if (s > max_s) s = max_s;
    if (s > 32) s = 32;
    return (int)(s < 1 ? 1 : s);
}

}  // namespace

extern "C" {

/* Scratch of cc_wgrad_tn_f16: the S partial copies of dW (nothing when one slice covers M). */
size_t cc_wgrad_tn_workspace_bytes(int32_t M, int32_t N1, int32_t N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0 || (N1 % WG_BN) || (N2 % WG_BN)) return 0;
    const int S = wgrad_slices(M, N1, N2);
    return S > 1 ? (size_t)S * N1 * N2 * 4 : 256;
}

int cc_wgrad_tn_f16(const void* dy_f16, const void* x_f16, float* dw, int32_t M, int32_t N1, int32_t N2,
                    const float* scale_dev, const float* col_partial, int32_t col_chunks, float* bias_grad, void* ws,
                    size_t ws_bytes, void* stream) {
    if (!dy_f16 || !x_f16 || !dw || M <= 0 || N1 <= 0 || N2 <= 0) return CC_ERR_INVALID;
    if ((N1 % WG_BN) || (N2 % WG_BN)) return CC_ERR_UNSUPPORTED;
    if ((col_partial != nullptr) != (bias_grad != nullptr) || (col_partial && col_chunks <= 0)) return CC_ERR_INVALID;
    const size_t need = cc_wgrad_tn_workspace_bytes(M, N1, N2);
    if (!ws || ws_bytes < need) return CC_ERR_WORKSPACE;
    WgradArgs a{};
    a.dy = static_cast<const _Float16*>(dy_f16);
    a.x = static_cast<const _Float16*>(x_f16);
    a.dw = dw;
    a.scale = scale_dev;
    a.M = M; a.N1 = N1; a.N2 = N2;
    a.S = wgrad_slices(M, N1, N2);
    const int steps = (M + WG_BK - 1) / WG_BK;
    a.rows_per_slice = (steps + a.S - 1) / a.S * WG_BK;
    a.partial = static_cast<float*>(ws);
    const size_t tiles = (size_t)(N1 / WG_BN) * (N2 / WG_BN);
    constexpr int smem = WG_NST * 2 * WG_TILE_BYTES;
    // (set per call: the attribute is per device, the call is cheap and idempotent - a process-wide "configured" flag would
    //  leave every device but the first one at the 64 KB default)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return CC_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(wgrad_tn_kernel, dim3((unsigned)(tiles * a.S)), dim3(256), smem, st, a);
    CC_LAUNCH_CHECK();
    if (a.S > 1 || col_partial) {
        const int64_t n4 = (int64_t)N1 * N2 / 4;
        const int main_blocks = a.S > 1 ? (int)((n4 + 255) / 256) : 0;
        const int bias_blocks = col_partial ? (N1 + 31) / 32 : 0;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)(main_blocks + bias_blocks)), dim3(256), 0, st, a.partial, dw, n4, a.S,
                           scale_dev, main_blocks, col_partial, col_chunks, N1, bias_grad);
        CC_LAUNCH_CHECK();
    }
    return CC_OK;
}

}  // extern "C"
