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
#include <hip/hip_ext.h>
#include <cstring>
#include <type_traits>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BK 64
// LDS stage buffers of a tile: three where they fit (the 8-wave 256x128 / 128x256 tiles: 3 x 48 KB) - the LDS-DMA loads of
// a stage are then issued TWO k-steps before its barrier instead of one, which is what an operand that comes from HBM
// rather than the Infinity Cache needs (a launch inside the step reads what the launch before it has just written: c_proj
// 55 us with its A operand left in the MALL by the previous launch of a loop, 65-75 us with it cold, tools/cold_operands.py).
#ifdef CC_TWO_STAGES
#define GEMM_NST(BM, BN, WAVES, BK) 2
#else
#define GEMM_NST(BM, BN, WAVES, BK) (((WAVES) == 8 && (BK) == 64 && 3 * ((BM) + (BN)) * (BK) * 2 <= 160 * 1024) ? 3 : 2)
#endif

// EPI_ATTN_LN epilogue LDS map (bytes): row statistics [256] x 8 | Q [256][72] | K [256][72] | V^T [64][328] (5 slots of 64
// keys or 8 of 32) | P strips 8 x [16][72] | sequence table [16] | V^T column of every tile row [256]
#define ATTN_QS 72
#define ATTN_VS 328
#define ATTN_Q_OFF 2048
#define ATTN_TAB_OFF (ATTN_Q_OFF + 2 * 256 * ATTN_QS * 2 + 64 * ATTN_VS * 2 + 8 * 16 * ATTN_QS * 2)
#define ATTN_SMEM (ATTN_TAB_OFF + 64 + 1024)

// the rows a residual epilogue adds: GemmArgs::R, or the output rows themselves (in place)
#ifdef CC_NO_RESID_SRC                      /* A/B: the round-4 form (always in place) */
#define CC_RESID_SRC(g) (reinterpret_cast<const float*>((g).C))
#else
#define CC_RESID_SRC(g) ((g).R ? (g).R : reinterpret_cast<const float*>((g).C))
#endif
__device__ __forceinline__ void glds16(const _Float16* g, _Float16* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// EPI_ATTN_LN: keys per V^T slot of a sequence of at most L tokens - 32 or 64 (short form: up to 8 / 5 sequences per row tile,
// every operand of an item in registers), else L rounded up to whole 32-key blocks (long form, L <= 256: ViT-B/16's 197-token
// frames, its 101-token clustered blocks, CLIP's native 77-token captions)
__host__ __device__ inline int attn_slot_keys(int L) { return L <= 32 ? 32 : (L <= 64 ? 64 : ((L + 31) & ~31)); }

__device__ __forceinline__ float quick_gelu(float x) {      // x * sigmoid(1.702 x), modules/clip.py:192-194
    // 5 VALU ops (v_exp_f32 + v_rcp_f32, ~1 ulp each): the epilogue of a 128x128 tile evaluates this
    // 64 times per lane, an IEEE divide + expf here costs more than the tile's MFMA work.
    const float e = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// development builds (-DCC_DEV_KNOBS) only: per-workgroup phase timestamps (entry, prologue done, loop done, exit)
#ifdef CC_DEV_KNOBS
__device__ long long* g_gemm_prof = nullptr;
#ifdef CC_STAMP_WALL
#define GEMM_CLOCK() wall_clock64()                       /* 100 MHz: a timeline in real time */
#else
#define GEMM_CLOCK() __builtin_readcyclecounter()
#endif
#define GEMM_PROF_INIT() long long* prof = g_gemm_prof
#define GEMM_STAMP(slot)                                                                                        \
    do {                                                                                                        \
        if (prof && threadIdx.x == 0) prof[(int64_t)blockIdx.x * 4 + (slot)] = (long long)GEMM_CLOCK(); \
    } while (0)
#else
#define GEMM_PROF_INIT() do { } while (0)
#define GEMM_STAMP(slot) do { } while (0)
#endif

#ifdef CC_DEV_KNOBS
// dev builds: the q | k | v values of the in_proj + attention form, written back as the two-launch form lays them out [M, 3W]
__device__ _Float16* g_attn_dump = nullptr;
extern "C" void cc_debug_set_attn_dump(_Float16* dev_buf) {
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dump), &dev_buf, sizeof(dev_buf));
}
#endif

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {          // lane exchange inside a 16-lane DPP row (bit pattern)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

// buffer descriptor over a whole output tensor: the epilogues' write-through (sc1) 16-byte stores go through it
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
}

template <int BM, int BN, int WM, int WN, int EPI, int BK = GEMM_BK>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f16_kernel(GemmPair pr) {
    GEMM_PROF_INIT();
    GEMM_STAMP(0);
    const bool second = (int)blockIdx.x >= pr.tiles0;
    GemmArgs g = second ? pr.p[1] : pr.p[0];
    if (g.m_dev) g.M = *g.m_dev;                             // device-side row count (compacted captions): wave-uniform
    constexpr int THREADS = 64 * WM * WN, NWAVES = WM * WN;
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;   // 16x16 fragments per wave (wave tile BM/WM x BN/WN)
    // (in_proj + attention on 224-row tiles - one 197-token ViT-B/16 frame: 14 fragment rows are multiplied, 256 rows staged)
    constexpr int BMS = (BM + 63) / 64 * 64;                 // rows of A staged per k-step
    constexpr int A_BYTES = BMS * BK * 2, B_BYTES = BN * BK * 2;
    constexpr int NST = GEMM_NST(BMS, BN, WM * WN, BK);
    constexpr int CH = BK / 8;                               // 16-byte chunks per staged row (8 or 16)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NST * (A_BYTES + B_BYTES)

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous run of
    // tiles (tn fastest) so tiles sharing an A panel hit the same L2.
    const int nwg = g.tiles_m * g.tiles_n;
    int bid = (int)blockIdx.x - (second ? pr.tiles0 : 0);
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // grouped rasterisation inside the XCD's run: the ~64 tiles an XCD has in flight form an
    // 8-row x 8-column patch, so both operand panels (8 A row-tiles + 8 W column-tiles) stay in its
    // 4 MiB L2 instead of streaming the whole weight matrix past it for every row of tiles.
    // (round 5 measured a group height of 1 for narrow outputs - whole rows of tiles per XCD, the A operand then crosses the
    // fabric 1.2 instead of 2.0 times - and found the launches SLOWER: profiles/r05_traffic_reconcile.txt; 8 stays)
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * g.tiles_n;
    const int group = bid / per_group, first_m = group * GROUP_M;
    const int gsz = min(g.tiles_m - first_m, GROUP_M);
    const int in_group = bid - group * per_group;
    const int tm = first_m + in_group % gsz, tn = in_group / gsz;
    // (EPI_ATTN_LN: a row tile is att_spt whole sequences, a column tile the q | k | v columns of head tn)
    constexpr bool ATTN = (EPI == EPI_ATTN_LN);
    static_assert(!ATTN || ((BM == 256 || BM == 224 || BM == 192) && BN == 192 && WM == 2 && WN == 4 && BK == 64), "in_proj + attention form");
    const int att_s0 = ATTN ? tm * g.att_spt : 0;
    const int row0 = ATTN ? (g.att_seq_off ? g.att_seq_off[att_s0] : att_s0 * g.att_L) : tm * BM, col0 = tn * BN;
    if (row0 >= g.M) return;                                 // (only with m_dev: the grid was sized for the upper bound)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WN, wc = wave % WN;

    // ---- staging addresses: LDS chunk idx -> (row r, chunk position cp); source chunk = cp ^ (r & (CH - 1))
    constexpr int A_LOADS = BMS * CH / THREADS, B_LOADS = BN * CH / THREADS;
    const _Float16* asrc[A_LOADS];
    const _Float16* bsrc[B_LOADS];
#pragma unroll
    for (int q = 0; q < A_LOADS; ++q) {
        const int idx = (q * NWAVES + wave) * 64 + lane, r = idx / CH, c = (idx % CH) ^ (r & (CH - 1));
        asrc[q] = g.A + (int64_t)min(row0 + r, g.M - 1) * (g.lda ? g.lda : g.K) + c * 8;
    }
#pragma unroll
    for (int q = 0; q < B_LOADS; ++q) {
        const int idx = (q * NWAVES + wave) * 64 + lane, r = idx / CH, c = (idx % CH) ^ (r & (CH - 1));
        // (ATTN: wave column wc multiplies the d-slice [16 wc, 16 wc + 16) of q, of k and of v: LDS row 48 wc + 16 sec + dd)
        const int wrow = ATTN ? ((r % 48) >> 4) * (g.N / 3) + tn * 64 + (r / 48) * 16 + (r & 15) : col0 + r;
        bsrc[q] = g.W + (int64_t)wrow * (g.ldw ? g.ldw : g.K) + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        _Float16* la = reinterpret_cast<_Float16*>(smem + buf * (A_BYTES + B_BYTES));
        _Float16* lb = reinterpret_cast<_Float16*>(smem + buf * (A_BYTES + B_BYTES) + A_BYTES);
#pragma unroll
        for (int q = 0; q < A_LOADS; ++q) glds16(asrc[q] + kt * BK, la + (q * NWAVES + wave) * 512);
#pragma unroll
        for (int q = 0; q < B_LOADS; ++q) glds16(bsrc[q] + kt * BK, lb + (q * NWAVES + wave) * 512);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = __builtin_amdgcn_readfirstlane(g.K / BK);
    // The rider's tiles are the ones that spill into a second round when the carrier alone fills the slots (the
    // clustered blocks): served first by the CU's arbiters they free their slots sooner for the tiles still queued.
    const bool rider_first = second && pr.rider_prio;
    if (rider_first) __builtin_amdgcn_s_setprio(2);
    stage(0, 0);
    // folded LayerNorm: thread r < BM reduces the producer's partial sums of tile row r right away (fixed
    // slot order, 8 loads in flight) - the L2 latency hides under the main loop; result parked in 2 registers.
    float row_mu = 0.f, row_rs = 1.f;
    if ((EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN || ATTN) && tid < BM) {
        const int m = min(row0 + tid, g.M - 1);
        const float2* ps = reinterpret_cast<const float2*>(g.ln_stats) + (int64_t)m * g.ln_slots;
        float sum = 0.f, sq = 0.f;
        for (int q0 = 0; q0 < g.ln_slots; q0 += 8) {
            float2 t2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t2[u] = (q0 + u < g.ln_slots) ? ps[q0 + u] : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) { sum += t2[u].x; sq += t2[u].y; }
        }
        row_mu = sum / (float)g.K;
        const float var = fmaxf(sq / (float)g.K - row_mu * row_mu, 0.f);
        row_rs = 1.0f / sqrtf(var + g.ln_eps);
    }
    // ATTN: thread r also finds where tile row r lands in the V^T tile (sequence slot * slot width + token), -1 = no sequence
    int att_vcol = -1;
    if (ATTN && tid < BM) {
        const int nst = min(g.att_spt, g.att_nseq - att_s0), slotw = attn_slot_keys(g.att_L);
        if (!g.att_seq_off) {
            const int sq = tid / g.att_L;
            if (sq < nst) att_vcol = sq * slotw + (tid - sq * g.att_L);
        } else {
            for (int j = 0; j < nst; ++j) {
                const int off = g.att_seq_off[att_s0 + j] - row0, len = g.att_seq_len[att_s0 + j];
                if (tid >= off && tid < off + len) att_vcol = j * slotw + (tid - off);
            }
        }
    }
    // residual epilogue with row centring: c_row = the row's mean one sublayer ago (the shift the previous producer used
    // + the mean of the centred copy it wrote), reduced here the same way; parked in row_mu until the epilogue
    if (EPI == EPI_F32_RESID_STATS && g.shift_stats && tid < BM) {
        const int m = min(row0 + tid, g.M - 1);
        const float2* ps = reinterpret_cast<const float2*>(g.shift_stats) + (int64_t)m * g.shift_slots;
        float sum = 0.f;
        for (int q0 = 0; q0 < g.shift_slots; q0 += 8) {
            float2 t2[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t2[u] = (q0 + u < g.shift_slots) ? ps[q0 + u] : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += t2[u].x;
        }
        row_mu = (g.shift_in ? g.shift_in[m] : 0.f) + sum / (float)g.N;
    }
    __syncthreads();
    GEMM_STAMP(1);
    const int l15 = lane & 15, lg = lane >> 4;
    // Per-column epilogue operands (bias, LN-fold column sums) are fetched into registers while the last k-step's
    // MFMAs run: issued from inside the epilogue, behind its stores, every fragment would pay its own L2 round trip.
    constexpr bool RESID = (EPI == EPI_F32_RESID || EPI == EPI_F32_RESID_STATS);
    constexpr bool LNFOLD = (EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN || ATTN);
    float4 biasv[NI], c1v[LNFOLD ? NI : 1];
    auto fetch_epilogue_operands = [&]() {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (ATTN && j == 2) {                       // the V fragment is accumulated transposed: one column (d) per lane
                const int n = 2 * (g.N / 3) + tn * 64 + wc * 16 + l15;
                biasv[j] = make_float4(g.bias[n], 0.f, 0.f, 0.f);
                c1v[LNFOLD ? j : 0] = make_float4(g.ln_c1[n], 0.f, 0.f, 0.f);
                continue;
            }
            const int n = ATTN ? j * (g.N / 3) + tn * 64 + wc * 16 + lg * 4 : col0 + wc * (BN / WN) + j * 16 + lg * 4;
            biasv[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (LNFOLD) c1v[j] = *reinterpret_cast<const float4*>(g.ln_c1 + n);
        }
    };
    // Main loop, half-shifted: the fragments of k-half 0 of step kt+1 are read while the MFMAs of k-half 1 of step
    // kt run, and those of k-half 1 while the MFMAs of k-half 0 run, so the one workgroup barrier per step sits
    // between two MFMA phases whose operands are already in registers (no LDS latency behind the barrier).
    //   step kt:  read F1(kt) | MFMA F0(kt) | barrier: stage kt+1 landed, buffer kt fully read
    //             | issue stage kt+2 into buffer kt | read F0(kt+1) | MFMA F1(kt)
    auto read_frags = [&](int buf, int ks, h8 (&af)[MI], h8 (&bf)[NI]) {
        const unsigned char* la = smem + buf * (A_BYTES + B_BYTES);
        const unsigned char* lb = la + A_BYTES;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = wr * (BM / WM) + i * 16 + l15;
            af[i] = *reinterpret_cast<const h8*>(la + r * (BK * 2) + (((ks * 4 + lg) ^ (r & (CH - 1))) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int r = wc * (BN / WN) + j * 16 + l15;
            bf[j] = *reinterpret_cast<const h8*>(lb + r * (BK * 2) + (((ks * 4 + lg) ^ (r & (CH - 1))) << 4));
        }
    };
    // Co-resident 4-wave workgroups: raising the wave priority over its MFMA block keeps the other workgroup's VMEM /
    // LDS instructions from being issued between them (measured: 1832 -> 1713-1763 cycles per k-step for the 128-wide
    // tiles; the single-workgroup 256x256 tile loses 7 % with it, so it is keyed on the loop form below).
        constexpr bool MMA_PRIO = !((NWAVES == 8) || (BM == 64 && BN == 64));
    auto mma = [&](const h8 (&af)[MI], const h8 (&bf)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // (ATTN: the V fragment with the operands swapped - a lane then holds 4 consecutive tile rows (keys) of one d,
                // the layout of the V^T tile the attention reads)
                if (ATTN && j == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
            }
    };
    auto read_a = [&](int buf, int ks, int i, h8 (&af)[MI]) {
        const unsigned char* la = smem + buf * (A_BYTES + B_BYTES);
        const int r = wr * (BM / WM) + i * 16 + l15;
        af[i] = *reinterpret_cast<const h8*>(la + r * (BK * 2) + (((ks * 4 + lg) ^ (r & (CH - 1))) << 4));
    };
    auto read_b = [&](int buf, int ks, int j, h8 (&bf)[NI]) {
        const unsigned char* lb = smem + buf * (A_BYTES + B_BYTES) + A_BYTES;
        const int r = wc * (BN / WN) + j * 16 + l15;
        bf[j] = *reinterpret_cast<const h8*>(lb + r * (BK * 2) + (((ks * 4 + lg) ^ (r & (CH - 1))) << 4));
    };
    auto stage_piece = [&](int buf, int kt, int q) {        // q-th of the A_LOADS + B_LOADS pieces of a stage
        unsigned char* base = smem + buf * (A_BYTES + B_BYTES);
        if (q < A_LOADS) glds16(asrc[q] + kt * BK, reinterpret_cast<_Float16*>(base) + (q * NWAVES + wave) * 512);
        else glds16(bsrc[q - A_LOADS] + kt * BK,
                    reinterpret_cast<_Float16*>(base + A_BYTES) + ((q - A_LOADS) * NWAVES + wave) * 512);
    };
    auto mma_row = [&](int i, const h8 (&af)[MI], const h8 (&bf)[NI]) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (ATTN && j == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
        }
    };
    // Measured (tools/gemm_phases.py, cycles per k-step): the half-shifted order wins where one workgroup owns the
    // CU (256x256: 3413 -> 2961) and for the 64x64 tile (1114 -> 953); with two 128-wide workgroups per CU the
    // plain order is faster (1832 vs 2033; re-measured in round 3 with the pinned phase-start wait: 1,468-1,572 vs 1,560-1,705)
    // - the co-resident workgroup already fills the LDS-latency gap.
        constexpr bool HALF_SHIFTED = (NWAVES == 8) || (BM == 64 && BN == 64);
    // residual rows requested from inside the three-stage loop (see there); the epilogue's row-major geometry
#ifdef CC_NO_RESID_PREFETCH
    constexpr bool RES_PREFETCH = false;
#else
    constexpr bool RES_PREFETCH = RESID && NST == 3 && HALF_SHIFTED && BK == GEMM_BK;
#endif
    constexpr int PF_LPRF = ((BN / WN) / 4 <= 8) ? 8 : 16, PF_RPP = 64 / PF_LPRF, PF_PASSES = 16 / PF_RPP;
    f32x4 resall[RES_PREFETCH ? MI : 1][RES_PREFETCH ? PF_PASSES : 1];
    constexpr bool res_have = RES_PREFETCH;
    if constexpr (BK > GEMM_BK) {
        // Deep k-step (BK = 128, the 64x64 tile only): the small tile is bound by the latency of the LDS-DMA round trip,
        // one per k-step and workgroup (~900 cycles for 8 MFMAs per wave at BK = 64) - twice the bytes per stage halves
        // the number of round trips of a long-K problem.  KS sub-steps of 32 k: the fragments of sub-step s+1 are read
        // while the MFMAs of sub-step s run; the next stage's loads are all issued behind the first reads.
        constexpr int KS = BK / 32, NLOAD = A_LOADS + B_LOADS;
        auto step = [&](int buf, int kt, bool with_stage) {
            h8 af[2][MI], bf[2][NI];
            read_frags(buf, 0, af[0], bf[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) read_frags(buf, ks + 1, af[(ks + 1) & 1], bf[(ks + 1) & 1]);
                if (ks == 0 && with_stage) {
#pragma unroll
                    for (int q = 0; q < NLOAD; ++q) stage_piece(buf ^ 1, kt + 1, q);
                }
                mma(af[ks & 1], bf[ks & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        for (int kt = 0; kt + 1 < nk; ++kt) step(kt & 1, kt, true);
        fetch_epilogue_operands();
        step((nk - 1) & 1, nk - 1, false);
    } else if constexpr (HALF_SHIFTED && NST == 3) {
        // The half-shifted loop over THREE stage buffers: stage kt + 3 is issued behind the barrier of step kt, which only
        // waits for stage kt + 1 - a counted s_waitcnt (the loads of stage kt + 2 stay in flight across it) and a raw
        // s_barrier: __syncthreads() would drain every LDS-DMA load in flight.  No other vector-memory operation is issued
        // between the prologue and the last step, so the count is exact.
        constexpr int NLOAD = A_LOADS + B_LOADS, LPG = (NLOAD + MI - 1) / MI, BPG = (NI + MI - 1) / MI;
        h8 a0[MI], b0[NI], a1[MI], b1[NI];
        if (nk > 1) stage(1, 1);
        if (nk > 2) stage(2, 2);
        read_frags(0, 0, a0, b0);
        // Residual epilogues: the tile's fp32 residual rows (128 KB per workgroup, 29.5 MB per launch from HBM - 4.3 us at the
        // HBM rate when every workgroup of the one round asks at the same moment, with every matrix core idle) are requested
        // from inside the loop, into registers in the epilogue's row-major layout: PF_G groups of PF_R 16-byte loads per lane,
        // one group behind the stage loads of each of the steps nk-10 .. nk-3.  Vector-memory operations return in order,
        // so a wait for stage kt + 1 also waits for everything issued before it - but not for what was issued after: the
        // waits of the last steps count the younger residual groups (the two of the steps kt-2 and kt-1) as allowed-
        // outstanding, and every group has two k-steps to arrive before anything waits for it.  Those steps are peeled
        // (tail_step<T>) so that the counts and the registers of a group are compile-time constants.
        // Measured (profiles/r04_resid_prefetch.txt): out_proj per workgroup 1.9 / 12.0 / 6.9 us (prologue / loop / epilogue)
        // -> 2.0 / 14.4 / 3.2, the launch alone 23.2 -> 19.9 us; the step 1.849 -> 1.823 ms in three same-session A/B rounds
        // (all 16 loads behind the last stage instead: 1.840).
        constexpr int PF_G = 8, PF_R = RES_PREFETCH ? MI * PF_PASSES / PF_G : 0;
        static_assert(!RES_PREFETCH || (MI * PF_PASSES == PF_G * PF_R && PF_PASSES % PF_R == 0), "residual groups");
        // (buffer loads: one per-lane byte offset for all groups, the group's row offset in an SGPR, rows behind M read as 0)
        __amdgpu_buffer_rsrc_t res_rsrc;
        int res_voff = 0;
        if constexpr (RES_PREFETCH) {
            const int64_t first = (int64_t)(row0 + wr * (BM / WM)) * g.ldc;            // first element of this wave's rows
            const int64_t left = ((int64_t)g.M * g.ldc - first) * 4;                    // bytes from there to the end of C
            res_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(CC_RESID_SRC(g)) + first, 0,
                                                         (int)(left < 0 ? 0 : (left > 0x7fffffff ? 0x7fffffff : left)), 0x00020000);
            res_voff = ((lane / PF_LPRF) * g.ldc + col0 + wc * (BN / WN) + (lane % PF_LPRF) * 4) * 4;
        }
        auto prefetch_group = [&](auto grp_c) {
            if constexpr (RES_PREFETCH) {
                constexpr int GRP = decltype(grp_c)::value;
#pragma unroll
                for (int u = 0; u < PF_R; ++u) {
                    constexpr int GPI = PF_PASSES / PF_R;             // groups per fragment row
                    const int i = GRP / GPI, ps = (GRP % GPI) * PF_R + u;
                    resall[i][ps] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                  res_rsrc, res_voff, (i * 16 + ps * PF_RPP) * g.ldc * 4, 0));
                }
            }
        };
        auto prefetch_all = [&]() {
            prefetch_group(std::integral_constant<int, 0>{}); prefetch_group(std::integral_constant<int, 1>{});
            prefetch_group(std::integral_constant<int, 2>{}); prefetch_group(std::integral_constant<int, 3>{});
            prefetch_group(std::integral_constant<int, 4>{}); prefetch_group(std::integral_constant<int, 5>{});
            prefetch_group(std::integral_constant<int, 6>{}); prefetch_group(std::integral_constant<int, 7>{});
        };
        const bool res_spread = RES_PREFETCH && nk >= PF_G + 2;   // (shorter K: requested before the loop - its waits then
        if (RES_PREFETCH && !res_spread) prefetch_all();          //  merely wait longer than they have to)
        auto phase1 = [&](int buf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mma_row(i, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BPG; ++q)
                    if (i * BPG + q < NI) read_b(buf, 1, i * BPG + q, b1);
                read_a(buf, 1, i, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto phase2 = [&](int buf, int nxt, int kt, bool with_stage) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (with_stage) {
#pragma unroll
                    for (int q = 0; q < LPG; ++q)
                        if (i * LPG + q < NLOAD) stage_piece(buf, kt + 3, i * LPG + q);
                }
                mma_row(i, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BPG; ++q)
                    if (i * BPG + q < NI) read_b(nxt, 0, i * BPG + q, b0);
                read_a(nxt, 0, i, a0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int cur = 0;
        auto plain_step = [&](int kt) {
            const int nxt = cur == 2 ? 0 : cur + 1;
            phase1(cur);
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");   // stage kt + 1 landed, kt + 2 in flight
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                         // ... for every wave; buffer `cur` fully read
            phase2(cur, nxt, kt, kt + 3 < nk);
            cur = nxt;
        };
        auto tail_step = [&](auto t_c) {                          // step kt = nk - 10 + T of the last PF_G + 1 steps
            constexpr int T = decltype(t_c)::value;
            const int kt = nk - (PF_G + 2) + T, nxt = cur == 2 ? 0 : cur + 1;
            phase1(cur);
            constexpr int younger = ((T - 2 >= 0 && T - 2 < PF_G) ? 1 : 0) + ((T - 1 >= 0 && T - 1 < PF_G) ? 1 : 0);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((T < PF_G ? NLOAD : 0) + younger * PF_R) : "memory");
            __builtin_amdgcn_s_barrier();
            phase2(cur, nxt, kt, T + 1 < PF_G);
            if constexpr (T < PF_G) prefetch_group(std::integral_constant<int, (T < PF_G ? T : 0)>{});
            cur = nxt;
        };
        if (res_spread) {
            for (int kt = 0; kt < nk - (PF_G + 2); ++kt) plain_step(kt);
            tail_step(std::integral_constant<int, 0>{}); tail_step(std::integral_constant<int, 1>{});
            tail_step(std::integral_constant<int, 2>{}); tail_step(std::integral_constant<int, 3>{});
            tail_step(std::integral_constant<int, 4>{}); tail_step(std::integral_constant<int, 5>{});
            tail_step(std::integral_constant<int, 6>{}); tail_step(std::integral_constant<int, 7>{});
            tail_step(std::integral_constant<int, 8>{});
        } else {
            for (int kt = 0; kt + 1 < nk; ++kt) plain_step(kt);
        }
        read_frags(cur, 1, a1, b1);
        mma(a0, b0);
        if constexpr (RES_PREFETCH) {                             // (__syncthreads() would drain the residual loads)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();                                      // every wave is done with the staging buffers
        }
        fetch_epilogue_operands();
        mma(a1, b1);
    } else if (HALF_SHIFTED) {
        // Every MFMA phase is written as MI groups of one fragment row (NI MFMAs), and the other work of the phase -
        // the fragment reads of the next k-half and, after the barrier, the LDS-DMA loads of the stage after next - is
        // dealt out over those groups in source order with a scheduling fence after each group: the issue cost of the
        // loads and reads hides under MFMA execution instead of preceding the block (in lockstep with the sibling wave
        // of the SIMD) or trailing it (straight into the barrier's wait).
        constexpr int NLOAD = A_LOADS + B_LOADS, LPG = (NLOAD + MI - 1) / MI;   // LDS-DMA pieces per group
        constexpr int BPG = (NI + MI - 1) / MI;                                 // B-fragment reads per group
        h8 a0[MI], b0[NI], a1[MI], b1[NI];
        if (nk > 1) stage(1, 1);
        read_frags(0, 0, a0, b0);
        // phase 1 of step kt: MFMAs of k-half 0, reads of k-half 1
        // The fragments a phase multiplies were read during the previous phase: wait for them BEFORE any read of this phase
        // is issued (hipcc otherwise hoists the first group's ds_reads above its own s_waitcnt lgkmcnt(0), which then also
        // waits for those fresh reads: a full LDS round trip of dead time at the head of every phase), and keep each group's
        // MFMAs ahead of its reads: 2,625 -> 2,468 cycles per k-step of the 256x256 tile.  In wall time the chip gives about
        // two thirds of that back as a lower clock (the big GEMMs run at the power limit): c_fc 53.6 -> 52.4 us, the step
        // 2.096 -> 2.085 ms over 6 same-session A/B rounds.
        auto phase1 = [&](int buf) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mma_row(i, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BPG; ++q)
                    if (i * BPG + q < NI) read_b(buf, 1, i * BPG + q, b1);
                read_a(buf, 1, i, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // phase 2: MFMAs of k-half 1, reads of k-half 0 of the next step (other buffer), loads of the stage after next
        auto phase2 = [&](int buf, int kt, bool with_stage) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                if (with_stage) {
#pragma unroll
                    for (int q = 0; q < LPG; ++q)
                        if (i * LPG + q < NLOAD) stage_piece(buf, kt + 2, i * LPG + q);
                }
                mma_row(i, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < BPG; ++q)
                    if (i * BPG + q < NI) read_b(buf ^ 1, 0, i * BPG + q, b0);
                read_a(buf ^ 1, 0, i, a0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int kt = 0;
        for (; kt + 2 < nk; ++kt) {                           // steady state: a stage to load in every step
            phase1(kt & 1);
            __syncthreads();
            phase2(kt & 1, kt, true);
        }
        for (; kt + 1 < nk; ++kt) {                           // second-to-last step: nothing left to load
            phase1(kt & 1);
            __syncthreads();
            phase2(kt & 1, kt, false);
        }
        read_frags((nk - 1) & 1, 1, a1, b1);
        mma(a0, b0);
        __syncthreads();                                      // every wave is done with the staging buffers
        fetch_epilogue_operands();
        mma(a1, b1);
    } else {
        // Two co-resident workgroups per CU: the plain order (this step's reads, then its MFMAs) with the same group
        // structure - the second k-half's reads and the next stage's LDS-DMA loads are dealt out between the fragment
        // rows of the first half's MFMAs; the wave priority is raised over the whole compute region.
        // (all LDS-DMA pieces go behind the FIRST fragment row: they have to land by this step's barrier, and
        // 2.196 vs 2.208 ms per step against dealing them out over the four rows)
        constexpr int NLOAD = A_LOADS + B_LOADS, LPG = NLOAD, BPG = (NI + MI - 1) / MI;
        auto step = [&](int buf, int kt, bool with_stage) {
            h8 a0[MI], b0[NI], a1[MI], b1[NI];
            if (MMA_PRIO) {                                   // (s_setprio ends a scheduling region: keep it outside)
                if (rider_first) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1);
            }
            read_frags(buf, 0, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                mma_row(i, a0, b0);
                if (with_stage) {
#pragma unroll
                    for (int q = 0; q < LPG; ++q)
                        if (i * LPG + q < NLOAD) stage_piece(buf ^ 1, kt + 1, i * LPG + q);
                }
#pragma unroll
                for (int q = 0; q < BPG; ++q)
                    if (i * BPG + q < NI) read_b(buf, 1, i * BPG + q, b1);
                read_a(buf, 1, i, a1);
                __builtin_amdgcn_sched_barrier(0);
            }
            mma(a1, b1);
            if (MMA_PRIO) {
                if (rider_first) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
        };
        for (int kt = 0; kt + 1 < nk; ++kt) step(kt & 1, kt, true);
        fetch_epilogue_operands();
        step((nk - 1) & 1, nk - 1, false);
    }

    GEMM_STAMP(2);
    // ---- epilogue: lane holds C[m = .. + l15][n = .. + lg*4 + 0..3]
    constexpr int WTM = BM / WM, WTN = BN / WN;              // wave tile
    constexpr bool OUT_F16 = (EPI == EPI_F16 || EPI == EPI_F16_GELU || EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN);
    constexpr bool FOLD_LN = (EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN);
    constexpr bool GELU = (EPI == EPI_F16_GELU || EPI == EPI_F16_GELU_LN);
    if constexpr (ATTN) {
        // ---- in_proj + attention.  The tile's q, k, v (fp16, the very values the two-launch form writes to HBM) go to LDS:
        // Q and K row-major [tile row][64 d] (both are the d-contiguous MFMA operands of S^T = K Q^T), V transposed
        // [64 d][sequence slot of 32 / 64 keys] (the key-contiguous operand of O^T = V^T P^T).  Then the (sequence, 16-query
        // tile) items of the tile are dealt over the 8 waves; each is the arithmetic of attention_wave_kernel (transformer.hip)
        // in the same order, so the two forms agree bit for bit.  q, k, v never reach HBM: 2 x 44 MB less traffic per block
        // at the bench shape, and one launch boundary less.
        constexpr int QS = ATTN_QS, VS = ATTN_VS, PS = ATTN_QS;
        float2* rowst = reinterpret_cast<float2*>(smem);
        _Float16* Qs = reinterpret_cast<_Float16*>(smem + ATTN_Q_OFF);
        _Float16* Ks = Qs + 256 * QS;
        _Float16* Vt = Ks + 256 * QS;
        _Float16* Pw = Vt + 64 * VS + wave * (16 * PS);
        int* stab = reinterpret_cast<int*>(smem + ATTN_TAB_OFF);   // [j]: first tile row of sequence j, [8 + j]: its length
        const int nst = min(g.att_spt, g.att_nseq - att_s0);
        const int slot = attn_slot_keys(g.att_L);
        // dev builds: -DCC_ATTN_STAMP_AT=n moves the third timeline stamp to point n of this epilogue, taken by thread
        // CC_ATTN_STAMP_TID (default 0; 448 = the first lane of a wave that writes V)
#if defined(CC_DEV_KNOBS) && defined(CC_ATTN_STAMP_AT)
#ifndef CC_ATTN_STAMP_TID
#define CC_ATTN_STAMP_TID 0
#endif
#define ATTN_STAMP_AT(n)                                                                                          \
    do {                                                                                                          \
        if ((n) == CC_ATTN_STAMP_AT && prof && threadIdx.x == CC_ATTN_STAMP_TID)                                  \
            prof[(int64_t)blockIdx.x * 4 + 2] = (long long)GEMM_CLOCK();                                          \
    } while (0)
#else
#define ATTN_STAMP_AT(n) do { } while (0)
#endif
        int* vcolS = stab + 16;                                    // [256]: V^T column of tile row r (prologue), -1 = none
        if (tid < BM) { rowst[tid] = make_float2(row_mu, row_rs); vcolS[tid] = att_vcol; }
        if (tid < 8) {
            const bool have = tid < nst;
            stab[tid] = !have ? (1 << 20) : (g.att_seq_off ? g.att_seq_off[att_s0 + tid] - row0 : tid * g.att_L);
            stab[8 + tid] = !have ? 0 : (g.att_seq_len ? g.att_seq_len[att_s0 + tid] : g.att_L);
        }
        __syncthreads();
        ATTN_STAMP_AT(1);
        {   // keys [len, slot) of every sequence's V slot are multiplied by P = 0: they must be finite
            const int d = tid & 63, sq = tid >> 6;
            if (sq < nst)
                for (int key = stab[8 + sq]; key < slot; ++key) Vt[d * VS + sq * slot + key] = (_Float16)0.f;
        }
        // Wave column wc holds the d-slice [16 wc, 16 wc + 16) of q (fragment 0), k (1) and v (2, transposed: a lane has rows
        // lg*4 .. +3 of column d = 16 wc + l15).  Rows (2a, 2a + 1) belong to one sequence when the sequences are uniform and
        // of even length: V^T then takes 4-byte writes (two keys of one d).
        // (one sequence per tile: tile row = key, any parity; the partner of the last key is a clamped row - finite, and P = 0 there)
        const bool pairs = !g.att_seq_off && ((g.att_L & 1) == 0 || g.att_spt == 1);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = wr * WTM + i * 16 + l15;
            {
                const float2 t2 = rowst[r];
                const float mu = t2.x, rs = t2.y;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x4 v = acc[i][j];
                    {   // (statement for statement the fold of the fp16-output epilogue below: the same contractions, the same bits)
                        const float4 c1 = c1v[j];
                        v[0] = rs * (v[0] - mu * c1.x); v[1] = rs * (v[1] - mu * c1.y);
                        v[2] = rs * (v[2] - mu * c1.z); v[3] = rs * (v[3] - mu * c1.w);
                    }
                    {
                        const float4 bb = biasv[j];
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                    *reinterpret_cast<h4*>((j == 0 ? Qs : Ks) + r * QS + wc * 16 + lg * 4) = o;
                }
            }
            {
                const int m0 = wr * WTM + i * 16 + lg * 4;         // this lane's 4 rows of the V fragment
                const float4 sa = *reinterpret_cast<const float4*>(rowst + m0), sb = *reinterpret_cast<const float4*>(rowst + m0 + 2);
                const int4 vc = *reinterpret_cast<const int4*>(vcolS + m0);
                const float c1 = c1v[2].x, bb = biasv[2].x;
                const f32x4 v = acc[i][2];
                // (two fused multiply-adds per value, as the compiler contracts the statements above)
                const _Float16 h0 = (_Float16)__builtin_fmaf(sa.y, __builtin_fmaf(-sa.x, c1, v[0]), bb);
                const _Float16 h1 = (_Float16)__builtin_fmaf(sa.w, __builtin_fmaf(-sa.z, c1, v[1]), bb);
                const _Float16 h2 = (_Float16)__builtin_fmaf(sb.y, __builtin_fmaf(-sb.x, c1, v[2]), bb);
                const _Float16 h3 = (_Float16)__builtin_fmaf(sb.w, __builtin_fmaf(-sb.z, c1, v[3]), bb);
                _Float16* vrow = Vt + (wc * 16 + l15) * VS;
                typedef _Float16 h2v __attribute__((ext_vector_type(2)));
                if (pairs) {
                    if (vc.x >= 0) *reinterpret_cast<h2v*>(vrow + vc.x) = h2v{h0, h1};
                    if (vc.z >= 0) *reinterpret_cast<h2v*>(vrow + vc.z) = h2v{h2, h3};
                } else {
                    if (vc.x >= 0) vrow[vc.x] = h0;
                    if (vc.y >= 0) vrow[vc.y] = h1;
                    if (vc.z >= 0) vrow[vc.z] = h2;
                    if (vc.w >= 0) vrow[vc.w] = h3;
                }
            }
        }
        ATTN_STAMP_AT(2);
        __syncthreads();
        ATTN_STAMP_AT(3);
#ifdef CC_DEV_KNOBS
        if (g_attn_dump) {
            const int Wd = g.N / 3;
            for (int idx = tid; idx < 256 * 64; idx += THREADS) {
                const int r = idx >> 6, d = idx & 63, vc = vcolS[r];
                if (vc >= 0) {
                    _Float16* o = g_attn_dump + (int64_t)(row0 + r) * g.N + tn * 64 + d;
                    o[0] = Qs[r * QS + d]; o[Wd] = Ks[r * QS + d]; o[2 * Wd] = Vt[d * VS + vc];
                }
            }
        }
#endif
        const int qtmax = (g.att_L + 15) >> 4;
        const bool CAUSAL = g.att_causal != 0;
        _Float16* Cb = reinterpret_cast<_Float16*>(g.C);
        // One item = (sequence, 16-query tile).  Straight-line per item: every LDS operand (Q, K, V^T fragments) is requested
        // up front, then the S MFMAs, the softmax, P through the wave's strip, the PV MFMAs - with the key-tile count a
        // template constant the compiler schedules the waits instead of paying a full LDS round trip per key tile.
        auto run_items = [&](auto nkt_c) {
            constexpr int NKT = decltype(nkt_c)::value, NKB = NKT / 2, SLOT = NKT * 16;
            for (int it = __builtin_amdgcn_readfirstlane(wave); it < nst * qtmax; it += NWAVES) {
                const int sq = it / qtmax, qt = it - sq * qtmax;   // wave-uniform
                const int off = stab[sq], L = stab[8 + sq];
                if (qt * 16 >= L) continue;
                const int q = qt * 16 + l15;
                h8 qf[2], kf[NKT][2], vf[NKB][4];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    qf[ks] = *reinterpret_cast<const h8*>(Qs + (off + min(q, L - 1)) * QS + (ks * 4 + lg) * 8);
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    const int kr = off + min(kt * 16 + l15, L - 1);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) kf[kt][ks] = *reinterpret_cast<const h8*>(Ks + kr * QS + (ks * 4 + lg) * 8);
                }
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
                        vf[kb][dt] = *reinterpret_cast<const h8*>(Vt + (dt * 16 + l15) * VS + sq * SLOT + (kb * 4 + lg) * 8);
                f32x4 sc[NKT];
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[kt][ks], qf[ks], a, 0, 0, 0);
                    sc[kt] = a;
                }
                float mx = -3.0e38f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    f32x4 a = sc[kt];
                    if (kt * 16 + 15 < L && (!CAUSAL || kt < qt)) { // wave-uniform: no key of this tile is masked
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            a[e] = a[e] * 0.125f;
                            mx = fmaxf(mx, a[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int key = kt * 16 + lg * 4 + e;
                            const bool ok = key < L && (!CAUSAL || key <= q);
                            a[e] = ok ? a[e] * 0.125f : -3.0e38f;
                            mx = fmaxf(mx, a[e]);
                        }
                    }
                    sc[kt] = a;
                }
                mx = cc_rows_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pexp = __expf(sc[kt][e] - mx);   // (a masked key: exp2 of -4e38 or -inf = 0 exactly)
                        sc[kt][e] = pexp;
                        sum += pexp;
                    }
                sum = cc_rows_sum(sum);
                const float inv = 1.0f / sum;
#pragma unroll
                for (int kt = 0; kt < NKT; ++kt) {
                    // (the product is rounded to fp32 and then to fp16, as in attention_wave_kernel: left alone, the compiler
                    // folds multiply and conversion into one v_fma_mixlo_f16 here - a single rounding, 1 ulp off in ~1e-5 of
                    // the entries)
                    float p0 = sc[kt][0] * inv, p1 = sc[kt][1] * inv, p2 = sc[kt][2] * inv, p3 = sc[kt][3] * inv;
                    asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
                    const h4 ph = {(_Float16)p0, (_Float16)p1, (_Float16)p2, (_Float16)p3};
                    *reinterpret_cast<h4*>(Pw + l15 * PS + kt * 16 + lg * 4) = ph;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
                f32x4 o[4];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    const h8 pf = *reinterpret_cast<const h8*>(Pw + l15 * PS + kb * 32 + lg * 8);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[kb][dt], pf, o[dt], 0, 0, 0);
                }
#ifndef CC_ATT_OUT_PLAIN
                // the 16 x 64 output tile goes back through the wave's strip (P has been consumed) so that a lane stores 16
                // bytes and 8 lanes a whole 128-byte row segment of one head - written through (sc1): nothing of it is left
                // dirty in the XCD's L2 for the write-back at the end of the launch (step 1.811 -> 1.802 ms in three same-
                // session rounds; the staging alone, with plain stores: 1.812)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const h4 oh = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                    *reinterpret_cast<h4*>(Pw + l15 * PS + dt * 16 + lg * 4) = oh;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qr = h * 8 + (lane >> 3), qq = qt * 16 + qr;
                    const h8 ov = *reinterpret_cast<const h8*>(Pw + qr * PS + (lane & 7) * 8);
                    if (qq < L) {
                        const int64_t e = (int64_t)(row0 + off + qq) * g.ldc + tn * 64 + (lane & 7) * 8;
                        if ((int64_t)g.M * g.ldc < (int64_t)0x3fffffff)      // (32-bit byte offset of the buffer form)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), wt_rsrc(Cb), (int)(e * 2), 0, 16);
                        else
                            *reinterpret_cast<h8*>(Cb + e) = ov;
                    }
                }
#else
                if (q < L) {
                    _Float16* dst = Cb + (int64_t)(row0 + off + q) * g.ldc + tn * 64;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const h4 oh = {(_Float16)o[dt][0], (_Float16)o[dt][1], (_Float16)o[dt][2], (_Float16)o[dt][3]};
                        *reinterpret_cast<h4*>(dst + dt * 16 + lg * 4) = oh;
                    }
                }
#endif
                __builtin_amdgcn_wave_barrier();
            }
        };
        // Long sequences (64 < slot <= 256 keys).  One item = (sequence, NQ adjacent 16-query tiles): the scores stay in registers
        // - up to NKTM key tiles of S^T = K Q^T per query tile, every K fragment read from LDS once for the NQ tiles - and so
        // does P: the accumulator layout of S^T (a lane holds keys kt*16 + lg*4 + 0..3 of query l15) IS an MFMA B operand of
        // O^T = V^T P^T once the 32 keys of a k-block are taken in the order [kt0: lg*4 + 0..3 | kt1: lg*4 + 0..3] - the
        // contraction over keys does not care about their order as long as the V^T fragment uses the same one (two 8-byte LDS
        // reads per fragment instead of one 16-byte read, shared by the NQ tiles).  No P strip, no wave barrier between the
        // softmax and the PV MFMAs; P = exp2((s - max) / 8 * log2 e) un-normalised in fp16 (<= 1), the row's 1 / sum scales the 16
        // output values instead of every P.  Two query tiles per item give a wave two independent dependency chains (197
        // tokens: 7 items on 8 waves in one round instead of 13 in two) - ViT-B/16 in_proj + attention 284 -> see
        // profiles/r05_forward_cfg5_kernel_stats.txt.  Causal sequences skip the key tiles behind the item's last query tile.
        // FULL: every sequence of the launch has exactly NKTM key tiles and no causal mask (the ViT's frames): the key-tile count
        // is a compile-time constant, the guards below fold away and the compiler pipelines the K / V fragment reads of an
        // item across key tiles instead of waiting for each guarded group (items of a 197-token frame 6.3 -> 4.7 us per workgroup, of two
        // 101-token frames 4.3 -> 3.1; cfg 5 6.05 -> 5.92 ms per step in three same-session rounds, profiles/r05_attention_two_query_tiles_ab.txt)
        auto run_items_long = [&](auto nktm_c, auto nq_c, auto full_c) {
            constexpr int NKTM = decltype(nktm_c)::value, NQ = decltype(nq_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            constexpr float SC = 0.125f * 1.4426950408889634f;     // 1 / sqrt(64) and the exp -> exp2 factor
            const int qgroups = (qtmax + NQ - 1) / NQ;
            for (int it = __builtin_amdgcn_readfirstlane(wave); it < nst * qgroups; it += NWAVES) {
                const int sq = it / qgroups, qt0 = (it - sq * qgroups) * NQ;   // wave-uniform
                const int off = stab[sq], L = stab[8 + sq];
                if (qt0 * 16 >= L) continue;
                int nkt = (((L + 15) >> 4) + 1) & ~1;               // key tiles of this sequence, whole 32-key blocks
                if (CAUSAL) nkt = min(nkt, (qt0 + NQ + 1) & ~1);
                nkt = FULL ? NKTM : __builtin_amdgcn_readfirstlane(min(nkt, NKTM));
                h8 qf[NQ][2];
#pragma unroll
                for (int u = 0; u < NQ; ++u)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        qf[u][ks] = *reinterpret_cast<const h8*>(Qs + (off + min((qt0 + u) * 16 + l15, L - 1)) * QS + (ks * 4 + lg) * 8);
                f32x4 sc[NQ][NKTM];
#pragma unroll
                for (int kt = 0; kt < NKTM; ++kt) {
#pragma unroll
                    for (int u = 0; u < NQ; ++u) sc[u][kt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (FULL || kt < nkt) {
                        const int kr = off + min(kt * 16 + l15, L - 1);
                        const h8 k0 = *reinterpret_cast<const h8*>(Ks + kr * QS + lg * 8);
                        const h8 k1 = *reinterpret_cast<const h8*>(Ks + kr * QS + (4 + lg) * 8);
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            const f32x4 a = __builtin_amdgcn_mfma_f32_16x16x32_f16(k0, qf[u][0], sc[u][kt], 0, 0, 0);
                            sc[u][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(k1, qf[u][1], a, 0, 0, 0);
                        }
                    }
                }
                float inv[NQ];
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int qt = qt0 + u, q = qt * 16 + l15;
                    float mx = -3.0e38f;
#pragma unroll
                    for (int kt = 0; kt < NKTM; ++kt) {
                        if (FULL || kt < nkt) {
                            f32x4 a = sc[u][kt];
                            if (!(kt * 16 + 15 < L && (!CAUSAL || kt < qt))) {   // (wave-uniform) a tile with masked keys
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    const int key = kt * 16 + lg * 4 + e;
                                    a[e] = (key < L && (!CAUSAL || key <= q)) ? a[e] : -3.0e38f;
                                }
                                sc[u][kt] = a;
                            }
                            mx = fmaxf(fmaxf(mx, a[0]), fmaxf(a[1], fmaxf(a[2], a[3])));
                        }
                    }
                    mx = cc_rows_max(mx);
                    const float mneg = -mx * SC;
                    float sum = 0.f;
#pragma unroll
                    for (int kt = 0; kt < NKTM; ++kt) {
                        if (FULL || kt < nkt) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float pexp = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[u][kt][e], SC, mneg));   // masked: exp2(-5e37) = 0
                                sc[u][kt][e] = pexp;
                                sum += pexp;
                            }
                        }
                    }
                    inv[u] = 1.0f / cc_rows_sum(sum);
                }
                f32x4 o[NQ][4];
#pragma unroll
                for (int u = 0; u < NQ; ++u)
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) o[u][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
                const _Float16* vbase = Vt + l15 * VS + sq * slot + lg * 4;
#pragma unroll
                for (int kb = 0; kb < NKTM / 2; ++kb) {
                    if (FULL || 2 * kb < nkt) {
                        h8 pf[NQ];
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            const f32x4 pa = sc[u][2 * kb], pb = sc[u][2 * kb + 1];
                            pf[u] = h8{(_Float16)pa[0], (_Float16)pa[1], (_Float16)pa[2], (_Float16)pa[3],
                                       (_Float16)pb[0], (_Float16)pb[1], (_Float16)pb[2], (_Float16)pb[3]};
                        }
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt) {
                            const h4 va = *reinterpret_cast<const h4*>(vbase + dt * 16 * VS + kb * 32);
                            const h4 vb = *reinterpret_cast<const h4*>(vbase + dt * 16 * VS + kb * 32 + 16);
                            const h8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
                            for (int u = 0; u < NQ; ++u) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u], o[u][dt], 0, 0, 0);
                        }
                    }
                }
                // each 16 x 64 output tile through the wave's strip -> 16-byte write-through stores (as the short form)
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int qt = qt0 + u;
                    if (qt * 16 >= L) break;                       // wave-uniform
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        const h4 oh = {(_Float16)(o[u][dt][0] * inv[u]), (_Float16)(o[u][dt][1] * inv[u]),
                                       (_Float16)(o[u][dt][2] * inv[u]), (_Float16)(o[u][dt][3] * inv[u])};
                        *reinterpret_cast<h4*>(Pw + l15 * PS + dt * 16 + lg * 4) = oh;
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int qr = h * 8 + (lane >> 3), qq = qt * 16 + qr;
                        const h8 ov = *reinterpret_cast<const h8*>(Pw + qr * PS + (lane & 7) * 8);
                        if (qq < L) {
                            const int64_t e = (int64_t)(row0 + off + qq) * g.ldc + tn * 64 + (lane & 7) * 8;
                            if ((int64_t)g.M * g.ldc < (int64_t)0x3fffffff)      // (32-bit byte offset of the buffer form)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), wt_rsrc(Cb), (int)(e * 2), 0, 16);
                            else
                                *reinterpret_cast<h8*>(Cb + e) = ov;
                        }
                    }
                    __builtin_amdgcn_s_waitcnt(0xc07f);            // the strip is rewritten by the next tile / item
                    __builtin_amdgcn_wave_barrier();
                }
            }
        };
        // (the lane's score registers: NQ x NKTM x 4 - two query tiles per item where that stays inside the wave's budget)
#ifdef CC_ATTN_LONG_NQ1                                        // A/B arm: one query tile per item
        constexpr int NQL = 1;
#else
        constexpr int NQL = 2;
#endif
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        const int nkt_all = (((g.att_L + 15) >> 4) + 1) & ~1;       // key tiles of a full-length sequence
#ifdef CC_ATTN_LONG_GUARDED                                    // A/B arm: always the guarded (run-time key-tile count) form
        const bool full_ok = false;
#else
        const bool full_ok = !CAUSAL && !g.att_seq_off;
#endif
        if (slot > 64 && full_ok && nkt_all == 14) run_items_long(integral_constant<int, 14>{}, integral_constant<int, NQL>{}, true_type{});
        else if (slot > 64 && full_ok && nkt_all == 8) run_items_long(integral_constant<int, 8>{}, integral_constant<int, NQL>{}, true_type{});
        else if (slot > 64 && full_ok && nkt_all == 12) run_items_long(integral_constant<int, 12>{}, integral_constant<int, NQL>{}, true_type{});
        else if (slot > 64 && full_ok && nkt_all == 10) run_items_long(integral_constant<int, 10>{}, integral_constant<int, NQL>{}, true_type{});
        else if (slot > 64 && full_ok && nkt_all == 6) run_items_long(integral_constant<int, 6>{}, integral_constant<int, NQL>{}, true_type{});
        else if (slot > 224) run_items_long(integral_constant<int, 16>{}, integral_constant<int, 1>{}, false_type{});
        else if (slot > 128) run_items_long(integral_constant<int, 14>{}, integral_constant<int, NQL>{}, false_type{});
        else if (slot > 64) run_items_long(integral_constant<int, 8>{}, integral_constant<int, NQL>{}, false_type{});
        else if (slot == 64) run_items(std::integral_constant<int, 4>{});
        else run_items(std::integral_constant<int, 2>{});
        GEMM_STAMP(3);
        return;
    }
    if (OUT_F16) {
        // fp16 outputs go through LDS (the staging buffers are free now) so that a wave stores whole
        // 128-byte row segments (8 lanes x 16 B) instead of 16 rows x 32 B per instruction.
        constexpr int LDO = WTN + 8;                          // halfs per staged row (144-byte rows for WTN = 64)
        constexpr int STATS_BYTES = FOLD_LN ? BM * 8 : 0;     // (mu, rstd) per tile row, at the start of smem
        constexpr int RH_MAX = (NST * (A_BYTES + B_BYTES) - STATS_BYTES) / (NWAVES * LDO * 2);
        constexpr int RH = (RH_MAX >= WTM) ? WTM : (RH_MAX >= 64 ? 64 : (RH_MAX >= 32 ? 32 : 16));
        static_assert(WTM % RH == 0 && RH % 16 == 0, "epilogue staging geometry");
        float2* rowst = reinterpret_cast<float2*>(smem);
        if (FOLD_LN) {
            static_assert(BM <= THREADS, "one thread per tile row");
            if (tid < BM) rowst[tid] = make_float2(row_mu, row_rs);
            __syncthreads();
        }
        _Float16* stg = reinterpret_cast<_Float16*>(smem + STATS_BYTES) + wave * (RH * LDO);
        constexpr int LPR_ACT = WTN / 8;                      // lanes that carry a row (16 B each)
        constexpr int LPR = LPR_ACT <= 4 ? 4 : (LPR_ACT <= 8 ? 8 : 16);   // lane slots per row (48-wide wave tiles: 6 of 8)
        constexpr int RPI = 64 / LPR;                         // rows per store instruction
        const int lr = lane / LPR, lc = (lane % LPR) * 8;
        const bool lane_on = (LPR == LPR_ACT) || (lane % LPR) < LPR_ACT;
        _Float16* Cb = reinterpret_cast<_Float16*>(g.C);
#pragma unroll
        for (int h0 = 0; h0 < WTM; h0 += RH) {
#pragma unroll
            for (int i = h0 / 16; i < (h0 + RH) / 16; ++i) {
                float mu = 0.f, rs = 1.f;
                if (FOLD_LN) { const float2 t2 = rowst[wr * WTM + i * 16 + l15]; mu = t2.x; rs = t2.y; }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    f32x4 v = acc[i][j];
                    if (FOLD_LN) {
                        const float4 c1 = c1v[LNFOLD ? j : 0];
                        v[0] = rs * (v[0] - mu * c1.x); v[1] = rs * (v[1] - mu * c1.y);
                        v[2] = rs * (v[2] - mu * c1.z); v[3] = rs * (v[3] - mu * c1.w);
                    }
                    {
                        const float4 bb = biasv[j];
                        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                    }
                    if (GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = quick_gelu(v[e]);
                    }
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)v[e];
                    *reinterpret_cast<h4*>(stg + (i * 16 + l15 - h0) * LDO + j * 16 + lg * 4) = o;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the strip is private to this wave
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r0 = 0; r0 < RH; r0 += RPI) {
                const int m = row0 + wr * WTM + h0 + r0 + lr;
                if (lane_on && m < g.M) {
                    const h8 ov = *reinterpret_cast<const h8*>(stg + (r0 + lr) * LDO + lc);
                    // The c_fc output (59 MB at the bench shape, the largest tensor of a block) is written through (sc1): it
                    // leaves the XCD's L2 while the launch runs instead of in the write-back at its end, and nobody re-reads
                    // it from this L2.  Measured per launch in situ (profiles/r04_store_policy.txt): c_fc 57.9 -> 55.6 us; the
                    // same policy on the in_proj output costs that launch 3.5 us, so it is keyed on the epilogue.
#ifdef CC_PLAIN_F16_STORES
                    constexpr bool wt_out = false;
#else
                    const bool wt_out = GELU && (int64_t)g.M * g.ldc < (int64_t)0x3fffffff;   // (32-bit byte offset of the buffer form)
#endif
                    if (wt_out)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ov), wt_rsrc(Cb),
                                                               (int)(((int64_t)m * g.ldc + col0 + wc * WTN + lc) * 2), 0, 16);
                    else
                        *reinterpret_cast<h8*>(Cb + (int64_t)m * g.ldc + col0 + wc * WTN + lc) = ov;
                }
            }
            if (h0 + RH < WTM) {
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
        GEMM_STAMP(3);
        return;
    }
    constexpr bool STATS = (EPI == EPI_F32_RESID_STATS);
    // fp32 outputs go through a per-wave LDS strip too: a 16-row fragment group is written in accumulator layout
    // (16 rows x 16 B per lane) and read back row-major, so that every global access of this epilogue - residual
    // read, fp32 write, fp16 copy - covers a whole wave-tile row (WTN x 4 B = 256 B for the 128-wide tiles) instead of
    // 64-byte pieces of 16 different rows, and a row's LN statistics reduce over one DPP row of lanes.
    constexpr int LDF = WTN + 4;                                  // floats per staged row (272-byte rows for WTN = 64)
    constexpr int LPRF_ACT = WTN / 4;                             // lanes that carry a row, 16 B each (16, 12 or 8)
    constexpr int LPRF = LPRF_ACT <= 8 ? 8 : 16;                  // lane slots per row (48-wide wave tiles: 12 of 16)
    constexpr int RPP = 64 / LPRF, PASSES = 16 / RPP;             // rows per access, accesses per 16-row group
    static_assert(ATTN || OUT_F16 || LPRF_ACT == 16 || LPRF_ACT == 8 || (LPRF_ACT == 12 && EPI == EPI_F32), "fp32 epilogue geometry");
    const bool flane_on = (LPRF == LPRF_ACT) || (lane % LPRF) < LPRF_ACT;
    static_assert(NWAVES * 16 * LDF * 4 <= 2 * (A_BYTES + B_BYTES), "fp32 epilogue strip fits the staging buffers");
    float* fstg = reinterpret_cast<float*>(smem) + wave * (16 * LDF);
    float* rowsh = reinterpret_cast<float*>(smem) + NWAVES * (16 * LDF);       // [BM] per-row shift (STATS with centring)
    static_assert(OUT_F16 || (NWAVES * 16 * LDF + BM) * 4 <= 2 * (A_BYTES + B_BYTES), "row shifts fit behind the strips");
    const bool centred = STATS && g.shift_stats != nullptr;
    if (centred) {
        if (tid < BM) rowsh[tid] = row_mu;
        __syncthreads();
    }
    const int er = lane / LPRF, ec = (lane % LPRF) * 4;           // this lane's row (within a pass) and column
    const int ncol = col0 + wc * WTN + ec;
    auto out_row = [&](int i, int ps) { return row0 + wr * WTM + i * 16 + ps * RPP + er; };
    // The fp32 residual rows come from HBM: they are fetched one fragment-row group ahead (issued before this
    // group's stores - a load behind a store to the same buffer cannot be hoisted by the compiler).
    float4 resv[2][PASSES];
    auto fetch_residual = [&](int slot, int i) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int m = min(out_row(i, ps), g.M - 1);
            resv[slot][ps] = *reinterpret_cast<const float4*>(CC_RESID_SRC(g) + (int64_t)m * g.ldc + ncol);
        }
    };
    // Where the fp32 residual tile is fetched was measured four ways (profiles/r04_gemm_timeline.txt; 9600 x 768 x 768,
    // 256x128 tile: prologue / loop / epilogue per workgroup): here, one fragment-row group ahead of its use: 1.8 / 11.1 /
    // 6.4 us; as the start value of the accumulators: 6.1 / 11.1 / 2.1; two fragments behind each of the first k-steps'
    // LDS-DMA loads: 1.8 / 15.9 / 2.3 (loads return in order, so every step's wait for its stage also waits for the
    // HBM-latency loads issued before it); added into the accumulators at one k-step, the workgroups dealt over four such
    // steps: 3.3 / 14.8 / 2.2 and results that depend on the tile and the grid.  The 128 KB cost a workgroup ~4 us wherever
    // they stand (16 half-used cache lines per instruction in any register-layout form; whole rows through the LDS strip
    // here) - the fetch stays in the epilogue, where the sum order is the same for every tile.
    constexpr bool RESID_LOAD = RESID;
    if (RESID_LOAD && !res_have) fetch_residual(0, 0);
    static_assert(!RES_PREFETCH || (PF_LPRF == LPRF && PF_RPP == RPP && PF_PASSES == PASSES), "prefetch geometry = epilogue geometry");
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        if (RESID_LOAD && !res_have && i + 1 < MI) fetch_residual((i + 1) & 1, i + 1);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const float4 bb = biasv[j];
            float4 sv = make_float4(acc[i][j][0] + bb.x, acc[i][j][1] + bb.y, acc[i][j][2] + bb.z, acc[i][j][3] + bb.w);
            *reinterpret_cast<float4*>(fstg + l15 * LDF + j * 16 + lg * 4) = sv;
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0): the strip is private to this wave
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int m = flane_on ? out_row(i, ps) : g.M;         // (idle lane slots of a 48-wide wave tile store nothing)
            float4 v = *reinterpret_cast<const float4*>(fstg + (ps * RPP + er) * LDF + (flane_on ? ec : 0));
            if (RESID) {
                if (RESID_LOAD) {
                    float4 c;
                    if constexpr (RES_PREFETCH) { const f32x4 t = resall[i][ps]; c = make_float4(t[0], t[1], t[2], t[3]); }
                    else c = resv[i & 1][ps];
                    v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w;
                }
                // write-through (sc1): the rows leave the XCD's L2 while the launch runs instead of in the write-back at its
                // end (nobody re-reads them from this L2: the consumer is another launch); step 2.000 -> 1.990 ms in 3 A/B rounds
#ifdef CC_PLAIN_RESID_STORES
                constexpr bool wt_ok = false;
#else
                const bool wt_ok = (int64_t)g.M * g.ldc < (int64_t)0x1fffffff;     // (the buffer form takes a 32-bit byte offset)
#endif
                if (m < g.M) {
                    if (wt_ok)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{v.x, v.y, v.z, v.w}), wt_rsrc(g.C),
                                                               (int)(((int64_t)m * g.ldc + ncol) * 4), 0, 16);
                    else
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + ncol) = v;
                }
                if (STATS) {
                    // the consumer multiplies fp16(h - c_row): without the centring the rounding error of the copy scales
                    // with |mean| / sigma of the row (LayerNorm itself is shift invariant, so the consumer is unchanged)
                    const float cr = centred ? rowsh[wr * WTM + i * 16 + ps * RPP + er] : 0.f;
                    const h4 o = {(_Float16)(v.x - cr), (_Float16)(v.y - cr), (_Float16)(v.z - cr), (_Float16)(v.w - cr)};
                    if (m < g.M) *reinterpret_cast<h4*>(g.c16 + (int64_t)m * g.ldc + ncol) = o;
                    if (centred && g.shift_out && tn == 0 && wc == 0 && (lane % LPRF) == 0 && m < g.M) g.shift_out[m] = cr;
                    // statistics of what the consumer will actually multiply: the fp16-rounded row
                    const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
                    float psum = (q0 + q1) + (q2 + q3);
                    float psq = (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
                    // the LPRF lanes of a row sit in one DPP row: quad swaps, half mirror (8 lanes), mirror (16 lanes)
                    psum += dpp_f32<0xB1>(psum);
                    psq += dpp_f32<0xB1>(psq);
                    psum += dpp_f32<0x4E>(psum);
                    psq += dpp_f32<0x4E>(psq);
                    psum += dpp_f32<0x141>(psum);
                    psq += dpp_f32<0x141>(psq);
                    if (LPRF == 16) {
                        psum += dpp_f32<0x140>(psum);
                        psq += dpp_f32<0x140>(psq);
                    }
                    if ((lane % LPRF) == 0 && m < g.M) {          // one slot per (tile column, wave column)
                        const int slot = tn * WN + wc;
                        reinterpret_cast<float2*>(g.stats_out)[(int64_t)m * (g.tiles_n * WN) + slot] = make_float2(psum, psq);
                    }
                }
            } else if (m < g.M) {
                if (EPI == EPI_F32_PATCH) {
                    const int f = m / g.patch_n, tok = m - f * g.patch_n + 1;
                    const float4 pe = *reinterpret_cast<const float4*>(g.pos + (int64_t)tok * g.N + ncol);
                    float* dst = reinterpret_cast<float*>(g.C) + ((int64_t)f * (g.patch_n + 1) + tok) * g.ldc + ncol;
                    *reinterpret_cast<float4*>(dst) = make_float4(v.x + pe.x, v.y + pe.y, v.z + pe.z, v.w + pe.w);
                } else {
                    float sc = g.out_scale != 0.f ? g.out_scale : 1.f;
                    if (g.out_unscale_dev) sc /= *g.out_unscale_dev;      // power of two: exact
                    const int nv = g.n_valid > 0 ? g.n_valid : g.N;
                    float* dst = reinterpret_cast<float*>(g.C) + (int64_t)m * g.ldc + ncol;
                    if (ncol + 3 < nv && (g.ldc & 3) == 0) {
                        *reinterpret_cast<float4*>(dst) = make_float4(sc * v.x, sc * v.y, sc * v.z, sc * v.w);
                    } else {
                        if (ncol < nv) dst[0] = sc * v.x;
                        if (ncol + 1 < nv) dst[1] = sc * v.y;
                        if (ncol + 2 < nv) dst[2] = sc * v.z;
                        if (ncol + 3 < nv) dst[3] = sc * v.w;
                    }
                }
            }
        }
        if (i + 1 < MI) {                                         // the strip is rewritten by the next group
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    GEMM_STAMP(3);
}

#ifdef CC_DEV_KNOBS
extern "C" int cc_debug_set_gemm_profile(long long* p) {   // development builds only; p [workgroups, 4] int64 device memory or NULL
    return hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_prof), &p, sizeof(p)) == hipSuccess ? CC_OK : CC_ERR_HIP;
}
#endif

namespace {

// Diagnostics (cc_debug_gemm_timing_*, declared in the diagnostics section of the public header; process-wide, not
// thread-safe - a measurement aid for bench.py): while armed, every launch of gemm_f16_kernel goes through hipExtLaunchKernelGGL with a start and a stop
// event, which receive the dispatch's own begin / end timestamps (what rocprofv3's kernel trace reads too) - so the
// duration of a kernel symbol can be read IN SITU, inside the eagerly launched step between its real neighbours, instead
// of from a stand-alone loop, and without the 2.5 - 5 us an event record of its own adds around a launch (measured: two
// adjacent hipEventRecord calls are 5.3 us apart on this stream).  Never armed during graph capture.
struct GemmTiming {
    bool armed = false;
    int count = 0, cap = 0;
    hipEvent_t* ev = nullptr;          // [2 * cap]: kernel start, kernel stop
    int (*info)[12] = nullptr;         // BM, BN, WM, WN, EPI, BK, M0, N0, K0, M1, N1, K1
} g_timing;

}  // namespace
bool cc_gemm_timing_claim(const int rec12[12], hipEvent_t* start, hipEvent_t* stop) {
    if (!(g_timing.armed && g_timing.count < g_timing.cap)) return false;
    const int tid = g_timing.count++;
    memcpy(g_timing.info[tid], rec12, sizeof(int) * 12);
    *start = g_timing.ev[2 * tid];
    *stop = g_timing.ev[2 * tid + 1];
    return true;
}
namespace {

template <int BM, int BN, int WM, int WN, int EPI, int BK = GEMM_BK>
int launch_one(const GemmPair& pr, int total, hipStream_t st) {
    constexpr int BMS = (BM + 63) / 64 * 64;
    constexpr size_t smem_loop = (size_t)GEMM_NST(BMS, BN, WM * WN, BK) * (size_t)(BMS + BN) * BK * 2;
    constexpr size_t smem = (EPI == EPI_ATTN_LN && smem_loop < ATTN_SMEM) ? (size_t)ATTN_SMEM : smem_loop;
    auto kern = gemm_f16_kernel<BM, BN, WM, WN, EPI, BK>;
    if (smem > 64 * 1024) {
        static bool configured = false;      // per instantiation; benign race (idempotent call)
        if (!configured) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem) != hipSuccess)
                return CC_ERR_HIP;
            configured = true;
        }
    }
    const int tid = (g_timing.armed && g_timing.count < g_timing.cap) ? g_timing.count++ : -1;
    if (tid >= 0) {
        const bool two = total > pr.tiles0;
        const int rec[12] = {BM, BN, WM, WN, EPI, BK, pr.p[0].M, pr.p[0].N, pr.p[0].K,
                             two ? pr.p[1].M : 0, two ? pr.p[1].N : 0, two ? pr.p[1].K : 0};
        memcpy(g_timing.info[tid], rec, sizeof(rec));
        hipExtLaunchKernelGGL(kern, dim3(total), dim3(64 * WM * WN), (unsigned)smem, st, g_timing.ev[2 * tid],
                              g_timing.ev[2 * tid + 1], 0u, pr);
    } else {
        hipLaunchKernelGGL(kern, dim3(total), dim3(64 * WM * WN), smem, st, pr);
    }
    CC_LAUNCH_CHECK();
    return CC_OK;
}

template <int BM, int BN, int WM, int WN, int BK = GEMM_BK>
int launch_tile(GemmArgs g0, const GemmArgs* g1, int epi, hipStream_t st) {
    GemmPair pr{};
    g0.tiles_m = (g0.M + BM - 1) / BM;
    g0.tiles_n = g0.N / BN;
    pr.p[0] = g0;
    pr.tiles0 = g0.tiles_m * g0.tiles_n;
    int total = pr.tiles0;
    if (g1) {
        pr.p[1] = *g1;
        pr.p[1].tiles_m = (g1->M + BM - 1) / BM;
        pr.p[1].tiles_n = g1->N / BN;
        total += pr.p[1].tiles_m * pr.p[1].tiles_n;
        // single-round carriers (the clustered blocks): the rider's tiles spill into a second round, see the kernel
        // (2.172 vs 2.182 ms per step over 5 A/B rounds; dev builds: CC_RIDER_PRIO=0 switches it off, =2 applies it to every launch)
#ifdef CC_DEV_KNOBS
        static const int rp = [] { const char* e = getenv("CC_RIDER_PRIO"); return e ? atoi(e) : 1; }();
#else
        constexpr int rp = 1;
#endif
        pr.rider_prio = (rp == 2) || (rp == 1 && g0.M < 5000);
    } else {
        pr.p[1] = g0;
    }
    switch (epi) {
        case EPI_F16: return launch_one<BM, BN, WM, WN, EPI_F16, BK>(pr, total, st);
        case EPI_F16_GELU: return launch_one<BM, BN, WM, WN, EPI_F16_GELU, BK>(pr, total, st);
        case EPI_F32_RESID: return launch_one<BM, BN, WM, WN, EPI_F32_RESID, BK>(pr, total, st);
        case EPI_F32_PATCH: return launch_one<BM, BN, WM, WN, EPI_F32_PATCH, BK>(pr, total, st);
        case EPI_F32: return launch_one<BM, BN, WM, WN, EPI_F32, BK>(pr, total, st);
        case EPI_F16_LN: return launch_one<BM, BN, WM, WN, EPI_F16_LN, BK>(pr, total, st);
        case EPI_F16_GELU_LN: return launch_one<BM, BN, WM, WN, EPI_F16_GELU_LN, BK>(pr, total, st);
        case EPI_F32_RESID_STATS: return launch_one<BM, BN, WM, WN, EPI_F32_RESID_STATS, BK>(pr, total, st);
        default: return CC_ERR_INVALID;
    }
}

// tiles whose wave tile is 48 columns wide exist for the fp16-output epilogues and the plain fp32 one
template <int BM, int BN, int WM, int WN>
int launch_tile_f16(GemmArgs g0, const GemmArgs* g1, int epi, hipStream_t st) {
    GemmPair pr{};
    g0.tiles_m = (g0.M + BM - 1) / BM;
    g0.tiles_n = g0.N / BN;
    pr.p[0] = g0;
    pr.tiles0 = g0.tiles_m * g0.tiles_n;
    int total = pr.tiles0;
    if (g1) {
        pr.p[1] = *g1;
        pr.p[1].tiles_m = (g1->M + BM - 1) / BM;
        pr.p[1].tiles_n = g1->N / BN;
        total += pr.p[1].tiles_m * pr.p[1].tiles_n;
        // single-round carriers (the clustered blocks): the rider's tiles spill into a second round, see the kernel
        // (2.172 vs 2.182 ms per step over 5 A/B rounds; dev builds: CC_RIDER_PRIO=0 switches it off, =2 applies it to every launch)
#ifdef CC_DEV_KNOBS
        static const int rp = [] { const char* e = getenv("CC_RIDER_PRIO"); return e ? atoi(e) : 1; }();
#else
        constexpr int rp = 1;
#endif
        pr.rider_prio = (rp == 2) || (rp == 1 && g0.M < 5000);
    } else {
        pr.p[1] = g0;
    }
    switch (epi) {
        case EPI_F16: return launch_one<BM, BN, WM, WN, EPI_F16>(pr, total, st);
        case EPI_F16_GELU: return launch_one<BM, BN, WM, WN, EPI_F16_GELU>(pr, total, st);
        case EPI_F16_LN: return launch_one<BM, BN, WM, WN, EPI_F16_LN>(pr, total, st);
        case EPI_F16_GELU_LN: return launch_one<BM, BN, WM, WN, EPI_F16_GELU_LN>(pr, total, st);
        case EPI_F32: return launch_one<BM, BN, WM, WN, EPI_F32>(pr, total, st);      // (the similarity GEMM)
        default: return CC_ERR_INVALID;
    }
}

}  // namespace

static bool gemm_shape_ok(const GemmArgs& g) {
    return g.M > 0 && g.N > 0 && g.K > 0 && (g.K % GEMM_BK) == 0 && (g.N % 64) == 0;
}

static bool epi_is_f16(int epi) {
    return epi == EPI_F16 || epi == EPI_F16_GELU || epi == EPI_F16_LN || epi == EPI_F16_GELU_LN;
}
static int tile_bn(int tile) { return (tile == 5 || tile == 10) ? 256 : tile == 7 ? 192 : (tile == 1 || tile == 3 || tile == 6) ? 128 : 64; }
static int tile_bk(int tile) { return tile == 8 ? 128 : GEMM_BK; }

// Residual epilogue (out_proj / c_proj) and the patch embedding at N % 128 == 0, M >= 4,800: the tile by a two-parameter time
// model per tile fitted to tools/resid_sweep.py on MI355X (gpurun_out/s2 of round 5 -> profiles/r05_resid_tile_sweep.txt):
//   launch = rounds x (a + b K) us,  rounds = whole rounds of the workgroup slots + (0.55 + 0.45 f) for a last round filled to f
// (a partly filled round runs faster than a full one - fewer workgroups share the L2 / fabric - but never in proportion);
// 128x128: 512 slots (two workgroups per CU), a = 9.1, b = 0.0182;  256x128: 256 slots, 9.3, 0.0170;  256x256: 256 slots, 17.2,
// 0.0281.  The fit reproduces the 24 measured launches (M = 3,200 ... 38,400, K = 768 / 3,072) within 10 %; what it changes
// against rounds 1-4: M = 12,800 (cfg 3's clustered blocks) 128x128 -> 256x256 (93 -> 74 us at K = 3,072), M = 25,600 (cfg 4)
// and M = 6,464 (cfg 5's clustered blocks) 128x128 -> 256x128 (176 -> 166, 51 -> 46 us); M = 25,600 then -> 128x256 (below).
// -> tile id, 0 = no opinion.
static int pick_resid_tile(int M, int N, int K) {
    if (M < 4800 || (N % 128)) return 0;
    auto rounds = [](long tiles, long slots) {
        const long full = tiles / slots, rest = tiles % slots;
        return (double)full + (rest ? 0.55 + 0.45 * (double)rest / (double)slots : 0.0);
    };
    const long m128 = (M + 127) / 128, m256 = (M + 255) / 256;
    double best = rounds(m128 * (N / 128), 512) * (9.1 + 0.0182 * K);
    int tile = 1;
    const double t6 = rounds(m256 * (N / 128), 256) * (9.3 + 0.0170 * K);
    if (t6 < best) { best = t6; tile = 6; }
    if ((N % 256) == 0) {
        const double t5 = rounds(m256 * (N / 256), 256) * (17.2 + 0.0281 * K);
        if (t5 < best) { best = t5; tile = 5; }
        // 128x256 (8 waves, three stage buffers): a = 11.3, b = 0.0141 - only where its grid is more than one round: as a single
        // round (M = 9,600: 225 tiles) it is 3-4 % faster than 256x128 stand-alone and 1.5 % SLOWER inside the step (cfg 2 1.886
        // -> 1.914 ms, three same-session rounds), at M = 25,600 (600 tiles) it wins both ways (cfg 4 4.429 -> 4.349 ms)
        const long t10n = m128 * (N / 256);
        const double t10 = rounds(t10n, 256) * (11.3 + 0.0141 * K);
#ifndef CC_NO_RESID_T10
        if (t10n > 256 && t10 < best) { best = t10; tile = 10; }
#endif
    }
    return tile;
}

static int pick_tile(const GemmArgs& g, int epi) {
    // measured on MI355X (tools/gemm_sweep.py): the 128x128 tile wins whenever it still yields
    // >= ~1.5 workgroups per CU; below that trade tile efficiency for parallelism.
    const long mt128 = (g.M + 127) / 128, mt64 = (g.M + 63) / 64;
    const bool n128 = (g.N % 128) == 0;
    const long t256 = (long)((g.M + 255) / 256) * (g.N / 256);
    // 256x256 (one 8-wave workgroup per CU) halves the L2->LDS bytes per flop; it only pays when the
    // tile count fills whole rounds of the 256 CUs
    const bool ok256 = (g.N % 256) == 0 && t256 >= 256 && (double)t256 / (double)(((t256 + 255) / 256) * 256) >= 0.65;
    // 256x192: the same 8-wave kernel with 48-column wave tiles.  Where the 256x256 grid leaves the last round of the
    // 256 CUs mostly idle (in_proj, N = 2304: 342 tiles = 1.34 rounds) the narrower tile fills the same number of rounds
    // with 3/4 of the work per round (456 tiles = 1.78 rounds).
    if (epi_is_f16(epi) && (g.N % 192) == 0) {
        const long t192 = (long)((g.M + 255) / 256) * (g.N / 192);
        const long r192 = (t192 + 255) / 256 * 192, r256 = ok256 ? (t256 + 255) / 256 * 256 : 0;
        if (t192 >= 256 && (double)t192 / (double)((t192 + 255) / 256 * 256) >= 0.65 && (!ok256 || r192 * 10 < r256 * 9))
            return 7;
    }
#ifndef CC_NO_RESID_TILE_MODEL
    if (epi == EPI_F32_RESID_STATS || epi == EPI_F32_RESID || epi == EPI_F32_PATCH) {
        const int t = pick_resid_tile(g.M, g.N, g.K);
        if (t) return t;
    }
#endif
    if (ok256) return 5;
    // residual epilogue at N = 768: the 256x128 tile (8 waves, one workgroup per CU) when its grid is one nearly full
    // round of the 256 CUs - half the A-panel re-reads of the 128x128 tile (stand-alone c_proj 55.5 vs 58.7 us = 816
    // TFLOP/s, out_proj 24.9 vs 26.2; on the step 2.124 vs 2.128 ms over 3 same-session A/B rounds)
    // (round 4: the same tile for the patch embedding - its A operand, the im2col matrix, is 58 MB the launch before has just
    // written, and the 256x128 tile runs on three stage buffers)
#ifdef CC_PATCH_TILE1
    const bool patch6 = false;
#else
    const bool patch6 = epi == EPI_F32_PATCH;
#endif
    if ((epi == EPI_F32_RESID_STATS || patch6) && n128) {
        const long t6 = (long)((g.M + 255) / 256) * (g.N / 128);
        if (t6 >= 200 && t6 <= 256) return 6;
    }
    if (n128 && mt128 * (g.N / 128) >= 300) return 1;
    if (n128 && mt64 * (g.N / 128) >= 400) return 3;
    if (mt128 * (g.N / 64) >= 400) return 2;
    // the small tile with 128-deep k-steps: c_proj of the clustered blocks 25.5 -> 21.2 us alone, 2.158 vs 2.166 ms per step;
    // since the round-3 loop (pinned phase-start wait) their out_proj (K = 768) gains too: 9.8 vs 10.7 us alone, 2.048-2.058
    // vs 2.059-2.061 ms per step in 3 same-session A/B rounds
    // (the backward's wgrad GEMMs - fp32 output, the row count as the contraction: the 128-deep tile holds one workgroup per CU
    // (96 KB of stage buffers), so a grid of 2.25 rounds runs three; the 64-deep one packs several per CU and does not care:
    // dW c_fc 3072 x 768 x 9600 80.9 vs 105.8 us, while dW in_proj (432 workgroups = 1.7 rounds) stays 58.8 vs 69.7)
    if (epi == EPI_F32) {
        const long wgs = mt64 * (g.N / 64);
        if (wgs > 256 && (double)wgs / (double)((wgs + 255) / 256 * 256) < 0.8) return 4;
    }
    return (g.K % 128 == 0 && g.K >= 768) ? 8 : 4;
}

// tile: 0 = auto, 1 = 128x128, 2 = 128x64, 3 = 64x128, 4 = 64x64 (4 waves); 5 = 256x256, 6 = 256x128, 7 = 256x192 (8 waves;
// 7 only for the fp16-output epilogues); 8 = 64x64 with 128-deep k-steps (K % 128 == 0); 10 = 128x256 (8 waves).
// (Ids 9 and 11 were the split-K and persistent stream-K forms of round 4: built, bit-checked, measured slower than the tiled
// kernel - profiles/r04_splitk.txt, r04_persist.txt - and removed in round 5.)
int cc_gemm_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, int tile, hipStream_t st, int* slots_out) {
    if (!gemm_shape_ok(g0) || (g1 && !gemm_shape_ok(*g1))) return CC_ERR_INVALID;
    if (tile == 0) {
        tile = pick_tile(g0, epi);
#ifdef CC_DEV_KNOBS
        // tuning aid (development builds only, -DCC_DEV_KNOBS): CC_TILE_E<epi>_<S|B>[_K<k>]=<tile> overrides the choice for
        // small (M < 5000) / big problems; the environment is scanned once
        static const bool any_override = [] {
            for (char** e = environ; e && *e; ++e)
                if (!strncmp(*e, "CC_TILE_", 8)) return true;
            return false;
        }();
        if (any_override) {
            char name[32];
            snprintf(name, sizeof(name), "CC_TILE_E%d_%c", epi, g0.M < 5000 ? 'S' : 'B');
            const char* ov = getenv(name);
            if (ov && atoi(ov) >= 1 && atoi(ov) <= 10 && atoi(ov) != 9) tile = atoi(ov);
            snprintf(name, sizeof(name), "CC_TILE_E%d_%c_K%d", epi, g0.M < 5000 ? 'S' : 'B', g0.K);   // one shape only
            ov = getenv(name);
            if (ov && atoi(ov) >= 1 && atoi(ov) <= 10 && atoi(ov) != 9) tile = atoi(ov);
        }
#endif
        if (g1) {                                  // the rider must be divisible by the carrier's BN
            if (g1->N % tile_bn(tile)) tile = (g1->N % 128 == 0 && (tile == 5 || tile == 7 || tile == 6)) ? 1 : 4;
            if (g1->K % tile_bk(tile)) tile = 4;
        }
    }
    if (tile == 7 && !epi_is_f16(epi) && epi != EPI_F32) return CC_ERR_INVALID;
    if ((g0.K % tile_bk(tile)) || (g1 && (g1->K % tile_bk(tile)))) return CC_ERR_INVALID;
    const int bn = tile_bn(tile);
    if ((g0.N % bn) || (g1 && (g1->N % bn))) return CC_ERR_INVALID;
    if (slots_out) {
        const int wn = (tile == 5 || tile == 10) ? 4 : 2;
        slots_out[0] = g0.N / bn * wn;
        slots_out[1] = g1 ? g1->N / bn * wn : 0;
        if (slots_out[0] > CC_LN_MAX_SLOTS || slots_out[1] > CC_LN_MAX_SLOTS) return CC_ERR_UNSUPPORTED;
    }
    switch (tile) {
        case 1: return launch_tile<128, 128, 2, 2>(g0, g1, epi, st);
        case 2: return launch_tile<128, 64, 2, 2>(g0, g1, epi, st);
        case 3: return launch_tile<64, 128, 2, 2>(g0, g1, epi, st);
        case 4: return launch_tile<64, 64, 2, 2>(g0, g1, epi, st);
        case 5: return launch_tile<256, 256, 2, 4>(g0, g1, epi, st);
        case 6: return launch_tile<256, 128, 4, 2>(g0, g1, epi, st);
        case 7: return launch_tile_f16<256, 192, 2, 4>(g0, g1, epi, st);
        case 8: return launch_tile<64, 64, 2, 2, 128>(g0, g1, epi, st);
        case 10: return launch_tile<128, 256, 2, 4>(g0, g1, epi, st);
        default: return CC_ERR_INVALID;
    }
}

// ------------------------------------------------------------------------------------------------ in_proj + attention
// One launch for q, k, v = in_proj(ln_1(x)) and softmax(q k^T / 8) v: a workgroup owns att_spt whole sequences x one head (256
// tile rows x its 192 weight rows), so the tile's q, k, v stay in LDS (EPI_ATTN_LN above).  Sequences per tile: as many as fit
// 256 rows and the V^T slots (5 of 64 keys, or 8 of 32 keys).
static bool attn_problem_ok(const GemmArgs& g) {
    return g.att_L > 0 && g.att_L <= 256 && g.att_nseq > 0 && g.N == 3 * g.K && (g.K % 64) == 0 && g.ldc == g.K && g.ln_stats &&
           g.ln_c1 && g.bias && !g.row_step && !g.row_map;
}
bool cc_gemm_attn_applies(const GemmArgs& g0, const GemmArgs* g1) {
#ifdef CC_NO_FUSED_ATTENTION
    return false;
#endif
    return attn_problem_ok(g0) && (!g1 || attn_problem_ok(*g1));
}
// Row-tile height of the launch: 256 rows, or 224 (one 197-token ViT-B/16 frame owns 14 fragment
// rows instead of 16: 1/8 less matrix-core work per tile) or 192 (the clustered blocks: 48 sequences of 50 tokens are 10
// tiles x 12 heads = 120 workgroups at 256 rows, 47 % of the CUs; 3 sequences per 192-row tile give 16 x 12 = 192 workgroups
// of 3/4 the work and two rounds of attention items per wave instead of three).  The choice minimises
// rounds x rows per tile over the whole launch (carrier + rider).
static int attn_spt(const GemmArgs& g, int bm) {
    // (V^T rows hold ATTN_VS - 8 = 320 keys; the sequence table 8 entries; L in (56, 64] takes the long form's path with 64-key slots)
    const int by_rows = bm / g.att_L, slots = (ATTN_VS - 8) / attn_slot_keys(g.att_L), by_slots = slots < 8 ? slots : 8;
    return by_rows < by_slots ? by_rows : by_slots;
}
static int attn_tile_rows(const GemmArgs& g0, const GemmArgs* g1) {
#ifdef CC_ATTN_BM256_ONLY                                      /* A/B arm: the round-5 form */
    return 256;
#else
    int best = 256;
    long best_cost = -1;
    const int cand[3] = {256, 224, 192};
    for (int bm : cand) {
        if (g0.att_L > bm || (g1 && g1->att_L > bm)) continue;
        const int s0 = attn_spt(g0, bm), s1 = g1 ? attn_spt(*g1, bm) : 1;
        if (s0 < 1 || s1 < 1) continue;
        long wgs = (long)((g0.att_nseq + s0 - 1) / s0) * (g0.K / 64);
        if (g1) wgs += (long)((g1->att_nseq + s1 - 1) / s1) * (g1->K / 64);
        const long cost = ((wgs + 255) / 256) * (bm + 96);     // rounds x (rows + the height-independent part of a tile)
        if (best_cost < 0 || cost < best_cost) { best = bm; best_cost = cost; }
    }
    return best;
#endif
}
int cc_gemm_attn_dispatch2(GemmArgs g0, const GemmArgs* g1, hipStream_t st) {
    if (!gemm_shape_ok(g0) || (g1 && !gemm_shape_ok(*g1)) || !cc_gemm_attn_applies(g0, g1)) return CC_ERR_INVALID;
    const int bm = attn_tile_rows(g0, g1);
    auto shape = [bm](GemmArgs& g) {
        g.att_spt = attn_spt(g, bm);
        g.tiles_m = (g.att_nseq + g.att_spt - 1) / g.att_spt;
        g.tiles_n = g.K / 64;
    };
    GemmPair pr{};
    shape(g0);
    pr.p[0] = g0;
    pr.tiles0 = g0.tiles_m * g0.tiles_n;
    int total = pr.tiles0;
    if (g1) {
        pr.p[1] = *g1;
        shape(pr.p[1]);
        total += pr.p[1].tiles_m * pr.p[1].tiles_n;
        pr.rider_prio = g0.M < 5000;
    } else {
        pr.p[1] = g0;
    }
    if (bm == 192) return launch_one<192, 192, 2, 4, EPI_ATTN_LN>(pr, total, st);
    if (bm == 224) return launch_one<224, 192, 2, 4, EPI_ATTN_LN>(pr, total, st);
    return launch_one<256, 192, 2, 4, EPI_ATTN_LN>(pr, total, st);
}

// ------------------------------------------------------------------------------------------------ selected rows
// The last block of a tower feeds only the rows its projection head reads (48 CLS rows of 2,400 at the bench shape):
// out_proj / c_fc / c_proj for those rows are weight-streaming problems (M <= 64 per chunk, 1.2 - 4.7 MB of weights),
// bound by the latency of the k chain, not by MFMA or HBM rate.  So: one workgroup per 32 output columns, its 8 waves
// split K (each wave keeps 4 k-steps = 24 16-byte loads per lane in flight, fragments loaded straight from global memory
// in MFMA layout - both operands are K-contiguous), partial tiles summed through LDS, and the epilogues of the main
// kernel (folded LayerNorm + QuickGELU; residual add with the centred fp16 copy + partial row statistics) applied on
// the physical rows.
struct RowsPair {
    GemmArgs p[2];
    int blocks0;
};

constexpr int ROWS_BM = 64, ROWS_BN = 32, ROWS_WAVES = 8, ROWS_LDR = ROWS_BN + 4;

template <int EPI>
__global__ __launch_bounds__(64 * ROWS_WAVES) void gemm_rows_kernel(RowsPair pr) {
    const bool second = (int)blockIdx.x >= pr.blocks0;
    const GemmArgs g = second ? pr.p[1] : pr.p[0];
    const int bid = second ? blockIdx.x - pr.blocks0 : blockIdx.x;
    const int ncb = g.N / ROWS_BN;
    const int cb = bid % ncb, row0 = (bid / ncb) * ROWS_BM, col0 = cb * ROWS_BN;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l15 = lane & 15, lg = lane >> 4;
    __shared__ float red[ROWS_WAVES / 2][ROWS_BM][ROWS_LDR];     // 36 KB
    __shared__ float2 rowst[ROWS_BM];
    constexpr bool LNFOLD = (EPI == EPI_F16_GELU_LN);
    constexpr bool STATS = (EPI == EPI_F32_RESID_STATS);
    // The kernel is a chain of memory round trips, so every load is issued as early as its address is known: (1) the
    // physical rows, (2) statistics + the first operand fragments + the epilogue operands, (3..) the rest of K.
    const int r = tid >> 3, sub = tid & 7, c = sub * 4;           // epilogue geometry: 8 threads per row, 4 columns each
    const int n = col0 + c;
    int mrow[5];                                                  // this lane's 4 fragment rows + its epilogue row
#pragma unroll
    for (int i = 0; i < 4; ++i) mrow[i] = min(row0 + i * 16 + l15, g.M - 1);
    mrow[4] = min(row0 + r, g.M - 1);
    int64_t prw[5];
    if (g.row_map) {
        int t[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) t[i] = g.row_map[mrow[i]];
#pragma unroll
        for (int i = 0; i < 5; ++i) prw[i] = t[i];
    } else {
        const int step = g.row_step ? g.row_step : 1;
#pragma unroll
        for (int i = 0; i < 5; ++i) prw[i] = (int64_t)mrow[i] * step;
    }
    const int64_t pm = prw[4];
    const bool on = row0 + r < g.M;
    // ---- this wave's K range
    const int tsteps = g.K / 32, spw = (tsteps + ROWS_WAVES - 1) / ROWS_WAVES;      // 32-wide k-steps, per wave
    const int kbeg = min(wave * spw, tsteps), ksteps = min(spw, tsteps - kbeg);        // (0 steps: a wave without work)
    const int klast = max(ksteps - 1, 0) * 32 + min(kbeg, tsteps - 1) * 32;            // a valid k offset for clamped loads
    const _Float16* wp[2];
    const _Float16* ap[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) wp[j] = g.W + (int64_t)(col0 + j * 16 + l15) * g.K + lg * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) ap[i] = g.A + prw[i] * g.K + lg * 8;
    // 3 k-steps = 18 16-byte loads per lane per batch, the next batch in flight under the MFMAs of the current one (6 steps
    // in one batch without the overlap measured slower: c_fc 12.5 vs 9.2 us)
    constexpr int U = 3;
    h8 bf[U][2], af[U][4];
    auto load_batch = [&](int s0) {                               // always U full steps: the ones behind the range re-read
#pragma unroll                                                    // the last valid step and multiply zeros
        for (int u = 0; u < U; ++u) {
            const int k = (s0 + u < ksteps) ? (kbeg + s0 + u) * 32 : klast;
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[u][j] = *reinterpret_cast<const h8*>(wp[j] + k);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[u][i] = *reinterpret_cast<const h8*>(ap[i] + k);
        }
    };
    load_batch(0);
    // per-row scalars (LayerNorm statistics / the centring shift): 8 threads per row take every 8th slot (<= 4 loads each)
    // and combine inside their 8-lane group
    float2 t2[CC_LN_MAX_SLOTS / 8];
    float shin = 0.f;
    const bool rowscal = LNFOLD || (STATS && g.shift_stats);
    if (rowscal) {
        const int nslots = LNFOLD ? g.ln_slots : g.shift_slots;
        const float2* ps = reinterpret_cast<const float2*>(LNFOLD ? g.ln_stats : g.shift_stats) + pm * nslots;
#pragma unroll
        for (int u = 0; u < CC_LN_MAX_SLOTS / 8; ++u) t2[u] = ps[min(sub + u * 8, nslots - 1)];
        if (!LNFOLD && g.shift_in) shin = g.shift_in[pm];
    }
    // epilogue operands
    const float4 bb = g.bias ? *reinterpret_cast<const float4*>(g.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 c1 = make_float4(0.f, 0.f, 0.f, 0.f), res = c1;
    if (LNFOLD) c1 = *reinterpret_cast<const float4*>(g.ln_c1 + n);
    else res = *reinterpret_cast<const float4*>(CC_RESID_SRC(g) + pm * g.ldc + n);
    if (rowscal) {
        const int nslots = LNFOLD ? g.ln_slots : g.shift_slots;
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int u = 0; u < CC_LN_MAX_SLOTS / 8; ++u) {
            const bool ok = sub + u * 8 < nslots;
            sum += ok ? t2[u].x : 0.f;
            sq += ok ? t2[u].y : 0.f;
        }
        sum += dpp_f32<0xB1>(sum);
        sq += dpp_f32<0xB1>(sq);
        sum += dpp_f32<0x4E>(sum);
        sq += dpp_f32<0x4E>(sq);
        sum += dpp_f32<0x141>(sum);
        sq += dpp_f32<0x141>(sq);
        if (sub == 0) {
            if (LNFOLD) {
                const float mu = sum / (float)g.K;
                const float var = fmaxf(sq / (float)g.K - mu * mu, 0.f);
                rowst[r] = make_float2(mu, 1.0f / sqrtf(var + g.ln_eps));
            } else {
                rowst[r] = make_float2(shin + sum / (float)g.N, 1.f);
            }
        }
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ksteps; s0 += U) {
        h8 bq[U][2], aq[U][4];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool ok = s0 + u < ksteps;
#pragma unroll
            for (int j = 0; j < 2; ++j) bq[u][j] = ok ? bf[u][j] : h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; ++i) aq[u][i] = af[u][i];
        }
        if (s0 + U < ksteps) load_batch(s0 + U);                 // next batch in flight under this one's MFMAs
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bq[u][j], aq[u][i], acc[i][j], 0, 0, 0);
    }
    // ---- sum the 8 partial tiles (the lane holds C[m = i*16 + l15][n = j*16 + lg*4 + 0..3]): the upper four waves park
    // theirs, the lower four add their own on top, every thread then sums four
    auto slot = [&](int i, int j) { return reinterpret_cast<float4*>(&red[wave & 3][i * 16 + l15][j * 16 + lg * 4]); };
    if (wave >= 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) *slot(i, j) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 t4 = *slot(i, j);
                *slot(i, j) = make_float4(acc[i][j][0] + t4.x, acc[i][j][1] + t4.y, acc[i][j][2] + t4.z, acc[i][j][3] + t4.w);
            }
    }
    __syncthreads();
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < ROWS_WAVES / 2; ++w) {
        const float4 t4 = *reinterpret_cast<const float4*>(&red[w][r][c]);
        v.x += t4.x; v.y += t4.y; v.z += t4.z; v.w += t4.w;
    }
    if (LNFOLD) {
        const float2 st = rowst[r];
        v.x = quick_gelu(st.y * (v.x - st.x * c1.x) + bb.x);
        v.y = quick_gelu(st.y * (v.y - st.x * c1.y) + bb.y);
        v.z = quick_gelu(st.y * (v.z - st.x * c1.z) + bb.z);
        v.w = quick_gelu(st.y * (v.w - st.x * c1.w) + bb.w);
        const h4 o = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
        if (on) *reinterpret_cast<h4*>(reinterpret_cast<_Float16*>(g.C) + pm * g.ldc + n) = o;
        return;
    }
    v.x += bb.x + res.x; v.y += bb.y + res.y; v.z += bb.z + res.z; v.w += bb.w + res.w;
    if (on) *reinterpret_cast<float4*>(reinterpret_cast<float*>(g.C) + pm * g.ldc + n) = v;
    if (STATS && g.c16) {
        const float cr = g.shift_stats ? rowst[r].x : 0.f;
        const h4 o = {(_Float16)(v.x - cr), (_Float16)(v.y - cr), (_Float16)(v.z - cr), (_Float16)(v.w - cr)};
        if (on) *reinterpret_cast<h4*>(g.c16 + pm * g.ldc + n) = o;
        if (on && g.shift_out && cb == 0 && c == 0) g.shift_out[pm] = cr;
        const float q0 = (float)o[0], q1 = (float)o[1], q2 = (float)o[2], q3 = (float)o[3];
        float psum = (q0 + q1) + (q2 + q3);
        float psq = (q0 * q0 + q1 * q1) + (q2 * q2 + q3 * q3);
        psum += dpp_f32<0xB1>(psum);                              // the 8 lanes of a row: quad swaps + half mirror
        psq += dpp_f32<0xB1>(psq);
        psum += dpp_f32<0x4E>(psum);
        psq += dpp_f32<0x4E>(psq);
        psum += dpp_f32<0x141>(psum);
        psq += dpp_f32<0x141>(psq);
        if (on && c == 0 && g.stats_out) reinterpret_cast<float2*>(g.stats_out)[pm * ncb + cb] = make_float2(psum, psq);
    }
}

bool cc_gemm_rows_ok(int N, int K, int epi) {
    if (N <= 0 || K <= 0 || (N % ROWS_BN) || (K % 32)) return false;
    if (epi == EPI_F32_RESID_STATS) return N / ROWS_BN <= CC_LN_MAX_SLOTS;
    return epi == EPI_F16_GELU_LN || epi == EPI_F32_RESID;
}

int cc_gemm_rows_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, hipStream_t st, int* slots_out) {
    if (!cc_gemm_rows_ok(g0.N, g0.K, epi) || (g1 && !cc_gemm_rows_ok(g1->N, g1->K, epi))) return CC_ERR_UNSUPPORTED;
    if (g0.M <= 0 || (g1 && g1->M <= 0)) return CC_ERR_INVALID;
    RowsPair pr{};
    pr.p[0] = g0;
    auto blocks = [](const GemmArgs& g) { return (g.N / ROWS_BN) * ((g.M + ROWS_BM - 1) / ROWS_BM); };
    pr.blocks0 = blocks(g0);
    int total = pr.blocks0;
    if (g1) { pr.p[1] = *g1; total += blocks(*g1); }
    if (slots_out) { slots_out[0] = g0.N / ROWS_BN; slots_out[1] = g1 ? g1->N / ROWS_BN : 0; }
    const dim3 grid(total), block(64 * ROWS_WAVES);
    switch (epi) {
        case EPI_F16_GELU_LN: hipLaunchKernelGGL(gemm_rows_kernel<EPI_F16_GELU_LN>, grid, block, 0, st, pr); break;
        case EPI_F32_RESID: hipLaunchKernelGGL(gemm_rows_kernel<EPI_F32_RESID>, grid, block, 0, st, pr); break;
        case EPI_F32_RESID_STATS: hipLaunchKernelGGL(gemm_rows_kernel<EPI_F32_RESID_STATS>, grid, block, 0, st, pr); break;
        default: return CC_ERR_UNSUPPORTED;
    }
    return hipGetLastError() == hipSuccess ? CC_OK : CC_ERR_HIP;
}

int cc_gemm_dispatch(GemmArgs g, int epi, int tile, hipStream_t st) { return cc_gemm_dispatch2(g, nullptr, epi, tile, st, nullptr); }

extern "C" {

int cc_linear_unscaled_f16(const void* a_f16, const void* w_f16, float* c, int32_t M, int32_t N, int32_t K,
                           const float* scale_dev, void* stream) {
    if (!a_f16 || !w_f16 || !c || !scale_dev) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(a_f16);
    g.W = static_cast<const _Float16*>(w_f16);
    g.C = c;
    g.M = M; g.N = N; g.K = K; g.ldc = N;
    g.out_unscale_dev = scale_dev;
    return cc_gemm_dispatch(g, EPI_F32, 0, static_cast<hipStream_t>(stream));
}
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

/* out [M, N] fp32 = resid + a w^T + bias: the residual epilogue of cc_linear_f16 ("f32_resid": out += ...) with the rows that
 * are added read from a tensor of their own (row stride N) - a forward that keeps its input for the backward needs no copy. */
int cc_linear_resid_f16(const void* a_f16, const void* w_f16, const float* bias, const float* resid, float* out, int32_t M,
                        int32_t N, int32_t K, int32_t tile, void* stream) {
    if (!a_f16 || !w_f16 || !resid || !out) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(a_f16);
    g.W = static_cast<const _Float16*>(w_f16);
    g.bias = bias;
    g.C = out;
    g.R = resid;
    g.M = M; g.N = N; g.K = K; g.ldc = N;
    return cc_gemm_dispatch(g, EPI_F32_RESID, tile, static_cast<hipStream_t>(stream));
}

/* LayerNorm-folded Linear: y = LN(h) W^T + b evaluated as rstd (h16 Wln^T - mu c1) + c2 (see cc_fold_layernorm_linear_f32).
 * h16 [M,K] fp16 copy of the rows, stats [M][slots][2] partial (sum, sum of squares) of those fp16 rows. */
int cc_linear_ln_f16(const void* h_f16, const void* w_ln_f16, const float* c1, const float* c2, const float* stats,
                     int32_t slots, float eps, void* out_f16, int32_t M, int32_t N, int32_t K, int32_t gelu,
                     int32_t tile, void* stream) {
    if (!h_f16 || !w_ln_f16 || !c1 || !c2 || !stats || !out_f16 || slots <= 0 || slots > CC_LN_MAX_SLOTS) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(h_f16);
    g.W = static_cast<const _Float16*>(w_ln_f16);
    g.bias = c2;
    g.C = out_f16;
    g.M = M; g.N = N; g.K = K; g.ldc = N;
    g.ln_stats = stats; g.ln_slots = slots; g.ln_c1 = c1; g.ln_eps = eps;
    return cc_gemm_dispatch(g, gelu ? EPI_F16_GELU_LN : EPI_F16_LN, tile, static_cast<hipStream_t>(stream));
}

/* in_proj with the LayerNorm folded + multi-head attention in ONE launch (modules/clip.py:210-214: ln_1 -> nn.MultiheadAttention's
 * in_proj and scaled-dot-product core): att[M, W] fp16 = softmax(q k^T / 8 [+ causal mask]) v per (sequence, head), with
 * q | k | v = LN(h) Wqkv^T + b evaluated as in cc_linear_ln_f16 (w_ln_f16 [3W, W], c1 / c2 [3W], stats [M][slots][2]).
 * nseq sequences of L tokens (row = s * L + t), or - seq_off / seq_len non-null - seq_len[s] <= L tokens from row seq_off[s]
 * (packed back to back); m_dev (may be null): device-side count of valid rows.  W = heads * 64, L <= 256.  The q, k, v rows of a
 * tile stay in LDS; for L <= 56 the results are bit-identical to cc_linear_ln_f16 followed by cc_attention_f16, for longer
 * sequences equal to the rounding of the fp16 output (the scores and P stay in registers, another key order inside the MFMAs).
 * CC_ERR_UNSUPPORTED when the shape is outside the fused form (the caller then runs the two launches). */
int cc_inproj_attention_f16(const void* h_f16, const void* w_ln_f16, const float* c1, const float* c2, const float* stats,
                            int32_t slots, float eps, void* att_f16, int32_t nseq, int32_t L, int32_t heads, int32_t causal,
                            const int32_t* seq_off, const int32_t* seq_len, const int32_t* m_dev, void* stream) {
    if (!h_f16 || !w_ln_f16 || !c1 || !c2 || !stats || !att_f16 || slots <= 0 || slots > CC_LN_MAX_SLOTS || nseq <= 0 || L <= 0 ||
        heads <= 0 || (seq_off == nullptr) != (seq_len == nullptr))
        return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(h_f16);
    g.W = static_cast<const _Float16*>(w_ln_f16);
    g.bias = c2;
    g.C = att_f16;
    g.M = nseq * L; g.K = heads * 64; g.N = 3 * g.K; g.ldc = g.K;
    g.ln_stats = stats; g.ln_slots = slots; g.ln_c1 = c1; g.ln_eps = eps;
    g.m_dev = m_dev;
    g.att_L = L; g.att_nseq = nseq; g.att_causal = causal; g.att_seq_off = seq_off; g.att_seq_len = seq_len;
    if (!cc_gemm_attn_applies(g, nullptr)) return CC_ERR_UNSUPPORTED;
    return cc_gemm_attn_dispatch2(g, nullptr, static_cast<hipStream_t>(stream));
}

/* Host-side query: the tile the dispatcher picks for a stand-alone launch of this shape and epilogue (CC_EPI_* or the
 * internal ids 5 = LN-folded f16, 6 = LN-folded f16 + QuickGELU, 7 = residual + statistics): 1 = 128x128, 2 = 128x64,
 * 3 = 64x128, 4 = 64x64 (4 waves), 5 = 256x256, 6 = 256x128, 7 = 256x192 (8 waves), 8 = 64x64 with 128-deep k-steps;
 * <= 0: unsupported.
 * (bench.py names the kernel instantiation a shape runs on with it.) */
int cc_linear_tile_for(int32_t M, int32_t N, int32_t K, int32_t epilogue) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K;
    if (!gemm_shape_ok(g) || epilogue < 0 || epilogue > EPI_F32_RESID_STATS) return CC_ERR_INVALID;
    return pick_tile(g, epilogue);
}

/* Host-side query: the number of partial-sum slots per row cc_linear_resid_stats_f16 writes for this shape and tile
 * (0 = auto), i.e. (N / tile columns) x (wave columns of the tile); <= 0: the shape / tile is not supported. */
int cc_linear_resid_stats_slots(int32_t M, int32_t N, int32_t K, int32_t tile) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K;
    if (!gemm_shape_ok(g)) return CC_ERR_INVALID;
    if (tile == 0) tile = pick_tile(g, EPI_F32_RESID_STATS);
    if (tile < 1 || tile > 10 || tile == 7 || tile == 9 || (K % tile_bk(tile)) || (N % tile_bn(tile))) return CC_ERR_INVALID;
    const int slots = N / tile_bn(tile) * ((tile == 5 || tile == 10) ? 4 : 2);
    return slots > CC_LN_MAX_SLOTS ? CC_ERR_UNSUPPORTED : slots;
}

/* Residual Linear that also emits what the next folded LayerNorm needs: h (fp32, in place) += a W^T + b;
 * h16 = fp16(h - c_row); stats_out [M][*slots_out][2] = per-tile partial (sum, sum of squares) of the fp16 rows;
 * c_row = shift_in[m] + mean of the previous centred copy (stats_in [M][slots_in][2]), written to shift_out [M]
 * (stats_in NULL: c_row = 0). */
int cc_linear_resid_stats_f16(const void* a_f16, const void* w_f16, const float* bias, float* h, void* h16_out,
                              float* stats_out, int32_t* slots_out, const float* shift_in, const float* stats_in,
                              int32_t slots_in, float* shift_out, int32_t M, int32_t N, int32_t K, int32_t tile,
                              void* stream) {
    if (!a_f16 || !w_f16 || !h || !h16_out || !stats_out || !slots_out) return CC_ERR_INVALID;
    if (stats_in && (slots_in <= 0 || slots_in > CC_LN_MAX_SLOTS || !shift_out)) return CC_ERR_INVALID;
    GemmArgs g{};
    g.A = static_cast<const _Float16*>(a_f16);
    g.W = static_cast<const _Float16*>(w_f16);
    g.bias = bias;
    g.C = h;
    g.M = M; g.N = N; g.K = K; g.ldc = N;
    g.c16 = static_cast<_Float16*>(h16_out);
    g.stats_out = stats_out;
    g.shift_in = shift_in; g.shift_stats = stats_in; g.shift_slots = slots_in; g.shift_out = shift_out;
    int slots[2] = {0, 0};
    const int rc = cc_gemm_dispatch2(g, nullptr, EPI_F32_RESID_STATS, tile, static_cast<hipStream_t>(stream), slots);
    *slots_out = slots[0];
    return rc;
}

int cc_debug_gemm_timing_begin(int cap) {
    if (g_timing.ev) {
        for (int i = 0; i < 2 * g_timing.cap; ++i) (void)hipEventDestroy(g_timing.ev[i]);
        delete[] g_timing.ev;
        delete[] g_timing.info;
        g_timing = GemmTiming{};
    }
    if (cap <= 0) return CC_OK;
    g_timing.ev = new hipEvent_t[2 * cap];
    g_timing.info = new int[cap][12];
    for (int i = 0; i < 2 * cap; ++i)
        if (hipEventCreate(&g_timing.ev[i]) != hipSuccess) return CC_ERR_HIP;
    g_timing.cap = cap;
    g_timing.armed = true;
    return CC_OK;
}

int cc_debug_gemm_timing_end(void) {
    g_timing.armed = false;
    return g_timing.count;
}

int cc_debug_gemm_timing_read(int i, float* us_out, int* info12_out) {
    if (i < 0 || i >= g_timing.count || !us_out || !info12_out) return CC_ERR_INVALID;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_timing.ev[2 * i], g_timing.ev[2 * i + 1]) != hipSuccess) return CC_ERR_HIP;
    *us_out = ms * 1e3f;
    memcpy(info12_out, g_timing.info[i], sizeof(int) * 12);
    return CC_OK;
}

}  // extern "C"
