// STATUS (round 4): built, bit-identical to the tiled kernel, measured SLOWER and therefore never picked by the dispatcher
// (tile 11 selects it; tests/test_r4_gpu.py keeps it correct).  c_fc 9600 x 3072 x 768: 110 us against 56 us tiled; with the
// schedule forced into scalar registers 72 - 88 us (and wrong results).  Per-workgroup stamps of that faster build, in_proj
// shape (profiles/r04_persist.txt): prologue 2.7 us, k-steps 1.6 - 2.0 us each (tiled: 1.57), handing a 256 KB partial over
// 2.7 - 6.7 us to send + ~4 us to fetch, epilogue 4.7 us (tiled: 2.6): the exchange and the slower tile ends cost a workgroup
// ~10 us where the equal shares save ~5.  What the tiled kernel loses to its ragged second round is cheaper than this cure.
//
// Persistent form of the 256x256 fp16 GEMM for the launches whose tile count is not a whole number of rounds of the
// 256 CUs (c_fc: 456 + 16 tiles = 1.84 rounds, in_proj: 456 tiles of 256x192 = 1.78; modules/clip.py:207-211,220-226).
//
//   * One workgroup per CU for the whole launch.  The k-steps of all tiles (carrier problem first, then the text rider's)
//     form one line of T steps; workgroup v - ranked XCD-major, so that an XCD's workgroups hold a contiguous run of
//     tiles - takes the steps [v T / G, (v + 1) T / G).  A range covers the TAIL of one tile, whole tiles, and the HEAD
//     of one more: every workgroup computes the same number of k-steps (+-1), there is no second round.
//   * The LDS-DMA pipeline never drains between tiles: the stage cursor runs two k-steps ahead of the compute cursor
//     across tile boundaries, so a tile's epilogue - which stages its stores through the 32 KB of LDS the two 64 KB
//     stage buffers leave free - has the next tile's first two k-steps in flight under it (no per-tile prologue).
//   * A workgroup walks its range from the END: the head of the last tile first, whole tiles, the tail of the first tile
//     last.  A tile cut between two workgroups: the holder of the head (k-steps 0 .. kc) computes it at time 0, writes its
//     accumulators in register layout (1 KB per fragment and wave: coalesced) to its exchange slot with write-through
//     stores and raises a flag behind the next workgroup barrier; the holder of the tail computes it LAST (it starts at
//     step T / G - (nk - kc) >= kc): its accumulators START from the slot, it runs k-steps kc .. nk and the epilogue.
//     The partial is therefore ready long before it is wanted - the flag is polled (bounded) for correctness, not for
//     time - and every output element is the same left-to-right sum over k as in the tiled kernel: the results are bit
//     for bit those of gemm_f16_kernel<256, 256>, whatever the cut.  Flags return to zero inside the launch.
//   * Hand-off rules of MI355X_MICROARCH.md: sc1 payload stores -> every wave s_waitcnt vmcnt(0) -> barrier -> relaxed
//     agent flag store; reader: relaxed agent poll -> barrier -> sc1 payload loads.  Correct wherever the two workgroups
//     run (nothing relies on the XCD placement, which only serves L2 locality of the operand panels).
//
// Same operands, swizzle, MFMA order (k ascending over the whole tile) and epilogue arithmetic as gemm_f16_kernel<256, 256>.
#include "cc_kernels.h"
#include <hip/hip_ext.h>
#include <type_traits>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int PBM = 256, PBN = 256, PBK = 64, PWN = 4, PTHREADS = 512, PWAVES = 8;
constexpr int PMI = 8, PNI = 4;                                   // 16x16 fragments per wave (wave tile 128 x 64)
constexpr int P_A_BYTES = PBM * PBK * 2, P_B_BYTES = PBN * PBK * 2, P_STAGE = P_A_BYTES + P_B_BYTES;
constexpr int P_SCR_OFF = 2 * P_STAGE, P_SCR_BYTES = 32 * 1024, P_SMEM = P_SCR_OFF + P_SCR_BYTES;   // 160 KB: the whole LDS
constexpr int PCH = PBK / 8;                                      // 16-byte chunks per staged row

struct PersistArgs {
    GemmArgs p[2];
    int tiles[2];            // 256x256 tiles per problem
    int nk[2];               // k-steps per tile
    int grid;                // workgroups (a multiple of 8)
    long long total_steps;   // tiles[0] * nk[0] + tiles[1] * nk[1]
    int* flags;              // [grid] zero before and after the launch
    unsigned char* slots;    // [grid] x 256 KB
    int* error;
};

__device__ __forceinline__ void glds16(const _Float16* g, unsigned char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ float quick_gelu(float x) {      // x * sigmoid(1.702 x), modules/clip.py:192-194
    const float e = __builtin_amdgcn_exp2f(-1.702f * 1.4426950408889634f * x);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t p_rsrc(void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
}

// a wave-uniform pointer / integer, told to the compiler (keeps the schedule's state in scalar registers: left to its own
// analysis hipcc carries the stage pointers per lane and spills LDS addresses to make room)
template <class Tp>
__device__ __forceinline__ Tp* uptr(Tp* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<Tp*>(((unsigned long long)hi << 32) | lo);
}

// development builds (-DCC_DEV_KNOBS) only: real-time stamps (100 MHz) per workgroup: [0] entry, [1] first stages landed,
// then per piece: k loop done, finish (epilogue / hand-over) done
#ifdef CC_DEV_KNOBS
__device__ long long* g_persist_prof = nullptr;
#define PERSIST_STAMP()                                                                                   \
    do {                                                                                                  \
        if (pprof && threadIdx.x == 0 && pslot < 16) pprof[(int64_t)blockIdx.x * 16 + pslot] = (long long)wall_clock64(); \
        ++pslot;                                                                                          \
    } while (0)
#define PERSIST_PROF_INIT() long long* pprof = g_persist_prof; int pslot = 0
#else
#define PERSIST_STAMP() do { } while (0)
#define PERSIST_PROF_INIT() do { } while (0)
#endif

struct Piece {               // one run of k-steps of one tile (all fields wave-uniform)
    int prob, tile, k0, k1, nk;
};

template <int EPI>
__global__ __launch_bounds__(PTHREADS) void gemm_persist_kernel(PersistArgs pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    PERSIST_PROF_INIT();
    PERSIST_STAMP();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / PWN, wc = wave % PWN, l15 = lane & 15, lg = lane >> 4;
    // (schedule integers: left to the compiler's own uniformity analysis.  Forcing them into scalar registers with
    // v_readfirstlane makes the loop 20 % faster - and the results wrong, 20 calls of 20: profiles/r04_persist.txt)
    auto uni = [](int x) { return x; };
    const int G = pa.grid;
    const int v = uni(((int)blockIdx.x & 7) * (G >> 3) + ((int)blockIdx.x >> 3));     // XCD-major rank
    const long long T = pa.total_steps;
    const int lo = uni((int)((long long)v * T / G)), hi = uni((int)((long long)(v + 1) * T / G));
    if (lo >= hi) return;
    const int steps0 = pa.tiles[0] * pa.nk[0];
    int m_prob[2] = {pa.p[0].M, pa.p[1].M};                       // (device-side row counts: compacted captions)
    if (pa.p[0].m_dev) m_prob[0] = uni(*pa.p[0].m_dev);
    if (pa.tiles[1] > 0 && pa.p[1].m_dev) m_prob[1] = uni(*pa.p[1].m_dev);

    auto piece_before = [&](int end) {                            // the piece that ENDS at line position end (lo < end <= hi)
        Piece pc;
        const int q = end - 1;
        pc.prob = q >= steps0 ? 1 : 0;
        const int rel = q - (pc.prob ? steps0 : 0);
        pc.nk = pc.prob ? pa.nk[1] : pa.nk[0];
        pc.tile = uni(rel / pc.nk);
        pc.k1 = uni(rel - pc.tile * pc.nk + 1);
        pc.k0 = uni(max(0, pc.k1 - (end - lo)));
        return pc;
    };
    auto tile_origin = [&](const Piece& pc, int& row0, int& col0) {   // grouped rasterisation (8 row tiles per group)
        const int tiles_m = pc.prob ? pa.p[1].tiles_m : pa.p[0].tiles_m, tiles_n = pc.prob ? pa.p[1].tiles_n : pa.p[0].tiles_n;
        const int per_group = 8 * tiles_n, group = pc.tile / per_group, first_m = group * 8;
        const int gsz = min(tiles_m - first_m, 8), in_group = pc.tile - group * per_group;
        row0 = uni((first_m + in_group % gsz) * PBM);
        col0 = uni((in_group / gsz) * PBN);
    };

    // ---------------------------------------------------------------- stage cursor (runs two k-steps ahead)
    const int srow = wave * 8 + (lane >> 3);                      // + q * 64: the tile row this lane stages in load q
    const int schunk = (lane & 7) ^ ((lane >> 3) & 7);            // source chunk (XOR swizzle through the SOURCE address)
    int s_end = hi, s_k = 0, s_k1 = 0, s_K = 0, s_issued = 0;   // the pieces in front of s_end are still to be staged
    const _Float16* s_a = nullptr;                                // A + row0 * K of the tile being staged
    const _Float16* s_w = nullptr;                                // W + col0 * K
    int offA[4], offW = 0;
    bool s_live = false;                                          // false: the staged tile has no rows (rider tile beyond *m_dev)
    auto stage_open = [&]() {                                     // the piece that ends at s_end is staged next
        const Piece pc = piece_before(s_end);
        s_end = uni(s_end - (pc.k1 - pc.k0));
        int row0, col0;
        tile_origin(pc, row0, col0);
        const int M = pc.prob ? m_prob[1] : m_prob[0];
        s_live = row0 < M;
        if (!s_live) return;                                      // (the operands of the last live stage stay valid)
        s_K = uni(pc.prob ? pa.p[1].K : pa.p[0].K);
        s_k = uni(pc.k0);
        s_k1 = uni(pc.k1);
        s_a = uptr((pc.prob ? pa.p[1].A : pa.p[0].A) + (int64_t)row0 * s_K);
        s_w = uptr((pc.prob ? pa.p[1].W : pa.p[0].W) + (int64_t)col0 * s_K);
#pragma unroll
        for (int q = 0; q < 4; ++q) offA[q] = (min(row0 + q * 64 + srow, M - 1) - row0) * s_K + schunk * 8;
        offW = srow * s_K + schunk * 8;
    };
    const _Float16* st_a = nullptr;                               // operands of the stage being issued (k offset applied)
    const _Float16* st_w = nullptr;
    unsigned char* st_lds = smem;
    auto stage_begin = [&]() {                                    // set up the next stage (buffer = its step index & 1)
        // (pieces of a rider tile without rows take no stage and no compute step: both cursors skip them)
        bool any = s_k < s_k1;
        while (!any && s_end > lo) {
            stage_open();
            any = s_live;
        }
        st_lds = smem + uni(s_issued & 1) * P_STAGE;
        s_issued = uni(s_issued + 1);
        // behind the end of the range the last stage is simply issued again (valid addresses, into a buffer nobody reads
        // any more): the MFMA phases then carry no branch around their LDS-DMA loads
        if (!any) return;
        st_a = uptr(s_a + s_k * PBK);
        st_w = uptr(s_w + s_k * PBK);
        s_k = uni(s_k + 1);
    };
    auto stage_piece = [&](int q) {                               // q-th of 8 LDS-DMA loads of the stage (A: 0..3, W: 4..7)
        if (q < 4) glds16(st_a + offA[q], st_lds + (q * PWAVES + wave) * 1024);
        else glds16(uptr(st_w + (int64_t)(q - 4) * 64 * s_K) + offW, st_lds + P_A_BYTES + ((q - 4) * PWAVES + wave) * 1024);
    };

    // ---------------------------------------------------------------- fragment reads / MFMA phases (gemm_f16_kernel's)
    f32x4 acc[PMI][PNI];
    auto read_a = [&](int buf, int ks, int i, h8 (&af)[PMI]) {
        const unsigned char* la = smem + buf * P_STAGE;
        const int r = wr * 128 + i * 16 + l15;
        af[i] = *reinterpret_cast<const h8*>(la + r * (PBK * 2) + (((ks * 4 + lg) ^ (r & (PCH - 1))) << 4));
    };
    auto read_b = [&](int buf, int ks, int j, h8 (&bf)[PNI]) {
        const unsigned char* lb = smem + buf * P_STAGE + P_A_BYTES;
        const int r = wc * 64 + j * 16 + l15;
        bf[j] = *reinterpret_cast<const h8*>(lb + r * (PBK * 2) + (((ks * 4 + lg) ^ (r & (PCH - 1))) << 4));
    };
    auto mma_row = [&](int i, const h8 (&af)[PMI], const h8 (&bf)[PNI]) {
#pragma unroll
        for (int j = 0; j < PNI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j], af[i], acc[i][j], 0, 0, 0);
    };
    h8 a0[PMI], b0[PNI], a1[PMI], b1[PNI];
    // phase 1 of a step: MFMAs of k-half 0 (fragments read during the previous phase), reads of k-half 1
    auto phase1 = [&](int buf) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < PMI; ++i) {
            mma_row(i, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < PNI) read_b(buf, 1, i, b1);
            read_a(buf, 1, i, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // phase 2: the loads of the stage after next into this step's buffer, MFMAs of k-half 1, and (unless the piece ends
    // here) the reads of k-half 0 of the next step from the other buffer
    auto phase2 = [&](int buf, bool read_next) {
#pragma unroll
        for (int i = 0; i < PMI; ++i) {
            stage_piece(i);
            mma_row(i, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            if (read_next) {
                if (i < PNI) read_b(buf ^ 1, 0, i, b0);
                read_a(buf ^ 1, 0, i, a0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---------------------------------------------------------------- prologue: the first two stages
    stage_begin();
    if (st_a == nullptr) return;                                  // nothing but rider tiles without rows in this range
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_piece(q);
    stage_begin();
#pragma unroll
    for (int q = 0; q < 8; ++q) stage_piece(q);
    __syncthreads();                                              // (vmcnt(0) + barrier: both landed)
    PERSIST_STAMP();

    constexpr bool OUT_F16 = (EPI == EPI_F16 || EPI == EPI_F16_GELU || EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN);
    constexpr bool FOLD_LN = (EPI == EPI_F16_LN || EPI == EPI_F16_GELU_LN);
    constexpr bool GELU = (EPI == EPI_F16_GELU || EPI == EPI_F16_GELU_LN);
    static_assert(OUT_F16, "persistent form: fp16-output epilogues");
    int n_epi = 0;                                                // epilogues run so far (selects the row-statistics set)
    int flag_pending = 0;                                         // a partial was stored: raise the flag behind the next barrier
    int gs = 0;                                                   // compute step index inside this workgroup's range
    int pos = hi;                                                 // the pieces in front of pos are still to be computed
    // One piece; its role is a compile-time constant (0: a whole tile; 1: the TAIL of a tile whose head the workgroup
    // ranked v - 1 holds - the accumulators start from its exchange slot, then the epilogue -; 2: the HEAD of a tile whose
    // tail the workgroup ranked v + 1 holds - the accumulators go to this workgroup's slot): three copies of the loop, each
    // without role branches.
    auto run_piece = [&](auto role_c, const Piece& pc, const GemmArgs& g, const int M, const int row0, const int col0) {
        constexpr int ROLE = decltype(role_c)::value;
        constexpr bool is_tail = ROLE == 1, is_head = ROLE == 2;
        const int n = uni(pc.k1 - pc.k0);
        // folded LayerNorm: thread r reduces the producer's partial sums of tile row r and parks (mu, rstd) in the spare LDS
        // (two sets, alternating per epilogue: a slow wave may still read the previous tile's while this one is written;
        // at least one workgroup barrier - the k loop's - separates a set's readers from its next writer)
        float2* rowst = reinterpret_cast<float2*>(smem + P_SCR_OFF) + (n_epi & 1) * PBM;
        if (FOLD_LN && !is_head && tid < PBM) {
            const int m = min(row0 + tid, M - 1);
            const float2* ps = reinterpret_cast<const float2*>(g.ln_stats) + (int64_t)m * g.ln_slots;
            float sum = 0.f, sq = 0.f;
            for (int q0 = 0; q0 < g.ln_slots; q0 += 8) {
                float2 t2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t2[u] = (q0 + u < g.ln_slots) ? ps[q0 + u] : make_float2(0.f, 0.f);
#pragma unroll
                for (int u = 0; u < 8; ++u) { sum += t2[u].x; sq += t2[u].y; }
            }
            const float mu = sum / (float)g.K;
            const float var = fmaxf(sq / (float)g.K - mu * mu, 0.f);
            rowst[tid] = make_float2(mu, 1.0f / sqrtf(var + g.ln_eps));
        }
        if (!is_head) ++n_epi;
        if constexpr (is_tail) {
            // ---- the accumulators start from the head's partial: the workgroup ranked v - 1 computed it first of all
            int* pflag = pa.flags + v - 1;
            if (tid == 0) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(4);
                    if (wall_clock64() - t0 > 20000000ll) { *pa.error = 1; break; }      // 0.2 s: never in a healthy run
                }
                __hip_atomic_store(pflag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the next launch starts from zeros
            }
            __syncthreads();
            const __amdgpu_buffer_rsrc_t rs = p_rsrc(pa.slots + (size_t)(v - 1) * CC_GEMM_SK_SLOT_BYTES);
            const int xoff = (wave * (PMI * PNI) * 64 + lane) * 16;
#pragma unroll
            for (int i = 0; i < PMI; ++i)
#pragma unroll
                for (int j = 0; j < PNI; ++j)
                    acc[i][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, xoff, (i * PNI + j) * 1024, 16));
        } else {
#pragma unroll
            for (int i = 0; i < PMI; ++i)
#pragma unroll
                for (int j = 0; j < PNI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- k loop of the piece (half-shifted: one barrier per step between two MFMA phases)
#pragma unroll
        for (int i = 0; i < PMI; ++i) {
            if (i < PNI) read_b(gs & 1, 0, i, b0);
            read_a(gs & 1, 0, i, a0);
        }
        auto step_head = [&]() {                                  // the part of a step in front of its second MFMA phase
            phase1(gs & 1);
            if (flag_pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (this wave's slot stores have left)
            __syncthreads();                                      // stage gs + 1 landed, buffer gs & 1 fully read
            if (flag_pending) {
                if (tid == 0) __hip_atomic_store(pa.flags + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag_pending = 0;
            }
            stage_begin();
        };
        for (int s = 0; s + 1 < n; ++s) {
            step_head();
            phase2(gs & 1, true);
            gs = uni(gs + 1);
        }
        // last step of the piece (peeled: the fragments of a next step are not read)
        step_head();
        phase2(gs & 1, false);
        gs = uni(gs + 1);
        PERSIST_STAMP();
        if constexpr (is_head) {
            // ---- hand the accumulators to the holder of the tile's tail
            const __amdgpu_buffer_rsrc_t rs = p_rsrc(pa.slots + (size_t)v * CC_GEMM_SK_SLOT_BYTES);
            const int xoff = (wave * (PMI * PNI) * 64 + lane) * 16;
#pragma unroll
            for (int i = 0; i < PMI; ++i)
#pragma unroll
                for (int j = 0; j < PNI; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs, xoff, (i * PNI + j) * 1024, 16);
            flag_pending = 1;
            if (pos <= lo) {                                      // nothing follows: publish now
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(pa.flags + v, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                flag_pending = 0;
            }
            PERSIST_STAMP();
            return;
        }
        // ---- epilogue through the spare LDS (the stage buffers hold the next tile's first k-steps)
        {
            constexpr int WTM = 128, WTN = 64, RH = 16, LDO = WTN + 8;       // 16-row strips of 144-byte rows per wave
            constexpr int STATS_BYTES = FOLD_LN ? 2 * PBM * 8 : 0;
            static_assert(STATS_BYTES + PWAVES * RH * LDO * 2 <= P_SCR_BYTES, "epilogue scratch");
            // (the row statistics were written in front of the k loop: its barriers lie between)
            _Float16* stg = reinterpret_cast<_Float16*>(smem + P_SCR_OFF + STATS_BYTES) + wave * (RH * LDO);
            // per-column epilogue operands: fetched here (the next tile's LDS-DMA loads are already in flight, so this L2
            // round trip costs the tile ~0.5 us once; held across the last MFMA phase they cost 32 registers the loop lacks)
            float4 biasv[PNI], c1v[FOLD_LN ? PNI : 1];
#pragma unroll
            for (int j = 0; j < PNI; ++j) {
                const int nn = col0 + wc * 64 + j * 16 + lg * 4;
                biasv[j] = g.bias ? *reinterpret_cast<const float4*>(g.bias + nn) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (FOLD_LN) c1v[j] = *reinterpret_cast<const float4*>(g.ln_c1 + nn);
            }
            const int lr = lane >> 3, lc = (lane & 7) * 8;        // 8 lanes carry a 128-byte row segment, 8 rows per store
            _Float16* Cb = reinterpret_cast<_Float16*>(g.C);
#pragma unroll
            for (int i = 0; i < PMI; ++i) {
                float mu = 0.f, rs = 1.f;
                if (FOLD_LN) { const float2 t2 = rowst[wr * WTM + i * 16 + l15]; mu = t2.x; rs = t2.y; }
#pragma unroll
                for (int j = 0; j < PNI; ++j) {
                    f32x4 vv = acc[i][j];
                    if (FOLD_LN) {
                        const float4 c1 = c1v[FOLD_LN ? j : 0];
                        vv[0] = rs * (vv[0] - mu * c1.x); vv[1] = rs * (vv[1] - mu * c1.y);
                        vv[2] = rs * (vv[2] - mu * c1.z); vv[3] = rs * (vv[3] - mu * c1.w);
                    }
                    const float4 bb = biasv[j];
                    vv[0] += bb.x; vv[1] += bb.y; vv[2] += bb.z; vv[3] += bb.w;
                    if (GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) vv[e] = quick_gelu(vv[e]);
                    }
                    h4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (_Float16)vv[e];
                    *reinterpret_cast<h4*>(stg + l15 * LDO + j * 16 + lg * 4) = o;
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the strip is private to this wave
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r0 = 0; r0 < RH; r0 += 8) {
                    const int m = row0 + wr * WTM + i * 16 + r0 + lr;
                    if (m < M)
                        *reinterpret_cast<h8*>(Cb + (int64_t)m * g.ldc + col0 + wc * WTN + lc) =
                            *reinterpret_cast<const h8*>(stg + (r0 + lr) * LDO + lc);
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
        PERSIST_STAMP();
    };
    while (pos > lo) {
        const Piece pc = piece_before(pos);
        int row0, col0;
        tile_origin(pc, row0, col0);
        const int M = pc.prob ? m_prob[1] : m_prob[0];
        pos = uni(pos - (pc.k1 - pc.k0));
        if (row0 >= M) continue;                                  // a rider tile without rows: nothing was staged for it
        const GemmArgs& g = pc.prob ? pa.p[1] : pa.p[0];
        if (pc.k0 > 0) run_piece(std::integral_constant<int, 1>{}, pc, g, M, row0, col0);
        else if (pc.k1 < pc.nk) run_piece(std::integral_constant<int, 2>{}, pc, g, M, row0, col0);
        else run_piece(std::integral_constant<int, 0>{}, pc, g, M, row0, col0);
    }
}

template <int EPI>
int launch_persist(const PersistArgs& pa, hipStream_t st) {
    auto kern = gemm_persist_kernel<EPI>;
    static bool configured = false;              // per instantiation; benign race (idempotent call)
    if (!configured) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P_SMEM) != hipSuccess)
            return CC_ERR_HIP;
        configured = true;
    }
    const bool two = pa.tiles[1] > 0;
    const int rec[12] = {PBM, PBN, 2, PWN, EPI, PBK | (3 << 16), pa.p[0].M, pa.p[0].N, pa.p[0].K,
                         two ? pa.p[1].M : 0, two ? pa.p[1].N : 0, two ? pa.p[1].K : 0};      // (3: the persistent form)
    hipEvent_t e0, e1;
    if (cc_gemm_timing_claim(rec, &e0, &e1))
        hipExtLaunchKernelGGL(kern, dim3(pa.grid), dim3(PTHREADS), (unsigned)P_SMEM, st, e0, e1, 0u, pa);
    else
        hipLaunchKernelGGL(kern, dim3(pa.grid), dim3(PTHREADS), P_SMEM, st, pa);
    CC_LAUNCH_CHECK();
    return CC_OK;
}

int g_cus = 0;

}  // namespace

#ifdef CC_DEV_KNOBS
extern "C" int cc_debug_set_persist_profile(long long* p) {   // development builds only; p [workgroups, 16] int64 device memory or NULL
    return hipMemcpyToSymbol(HIP_SYMBOL(g_persist_prof), &p, sizeof(p)) == hipSuccess ? CC_OK : CC_ERR_HIP;
}
#endif

// Does the persistent form apply?  fp16-output epilogue, both problems tileable by 256 columns, an exchange scratch, and at
// least one tile per workgroup (so that a workgroup's range covers a whole tile length: a tile is cut at most once).
bool cc_gemm_persist_applies(const GemmArgs& g0, const GemmArgs* g1, int epi) {
    if (!(epi == EPI_F16 || epi == EPI_F16_GELU || epi == EPI_F16_LN || epi == EPI_F16_GELU_LN)) return false;
    if (!g0.sk_ws || g0.row_step || g0.row_map) return false;
    auto ok = [](const GemmArgs& g) { return g.M > 0 && (g.N % 256) == 0 && (g.K % 64) == 0; };
    if (!ok(g0) || (g1 && !ok(*g1))) return false;
    if (g_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        const bool have = hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess;
        g_cus = have ? prop.multiProcessorCount : 256;
    }
    const int G = (g_cus < CC_GEMM_SK_MAX_SLOTS ? g_cus : CC_GEMM_SK_MAX_SLOTS) & ~7;
    auto tiles = [](const GemmArgs& g) { return (long long)((g.M + 255) / 256) * (g.N / 256); };
    const long long t0 = tiles(g0), t1 = g1 ? tiles(*g1) : 0;
    const long long T = t0 * (g0.K / 64) + (g1 ? t1 * (g1->K / 64) : 0);
    const int nkmax = g1 && g1->K > g0.K ? g1->K / 64 : g0.K / 64;
    if (G < 8 || T / G < nkmax) return false;                    // fewer k-steps per workgroup than one tile holds
    // worth it where the one-round-per-tile grid leaves a ragged last round: between 1.05 and 3.95 rounds
    const long long tt = t0 + t1;
    return tt * 100 >= (long long)G * 105 && tt * 100 <= (long long)G * 395;
}

int cc_gemm_persist_dispatch2(GemmArgs g0, const GemmArgs* g1, int epi, hipStream_t st) {
    if (!cc_gemm_persist_applies(g0, g1, epi)) return CC_ERR_UNSUPPORTED;
    PersistArgs pa{};
    auto prep = [](GemmArgs& g) { g.tiles_m = (g.M + 255) / 256; g.tiles_n = g.N / 256; };
    prep(g0);
    pa.p[0] = g0;
    pa.tiles[0] = g0.tiles_m * g0.tiles_n;
    pa.nk[0] = g0.K / 64;
    if (g1) {
        pa.p[1] = *g1;
        prep(pa.p[1]);
        pa.tiles[1] = pa.p[1].tiles_m * pa.p[1].tiles_n;
        pa.nk[1] = g1->K / 64;
    } else {
        pa.p[1] = g0;
        pa.tiles[1] = 0;
        pa.nk[1] = 1;
    }
    pa.grid = (g_cus < CC_GEMM_SK_MAX_SLOTS ? g_cus : CC_GEMM_SK_MAX_SLOTS) & ~7;
    pa.total_steps = (long long)pa.tiles[0] * pa.nk[0] + (long long)pa.tiles[1] * pa.nk[1];
    pa.flags = static_cast<int*>(g0.sk_ws);
    pa.error = pa.flags + CC_GEMM_SK_FLAG_BYTES / 4 - 1;
    pa.slots = static_cast<unsigned char*>(g0.sk_ws) + CC_GEMM_SK_FLAG_BYTES;
    switch (epi) {
        case EPI_F16: return launch_persist<EPI_F16>(pa, st);
        case EPI_F16_GELU: return launch_persist<EPI_F16_GELU>(pa, st);
        case EPI_F16_LN: return launch_persist<EPI_F16_LN>(pa, st);
        case EPI_F16_GELU_LN: return launch_persist<EPI_F16_GELU_LN>(pa, st);
        default: return CC_ERR_UNSUPPORTED;
    }
}
