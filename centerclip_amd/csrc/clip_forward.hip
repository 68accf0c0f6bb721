// Host-side orchestration of the CLIP visual / text forward: one C call enqueues the whole
// encoder on a HIP stream (no host synchronisation, no allocation - the workspace is caller
// owned), so it can be captured into a hipGraph as is.
//
// Activations are frame-major ([frame, token, width], "NLD"): attention works on contiguous
// per-frame rows and the token-cluster op reads/writes the same buffer through strides.
// Reference call stack: CLIP.encode_image -> VisualTransformer.forward -> Transformer ->
// ResidualAttentionBlock.forward (modules/clip.py:460-469, 304-349, 256-269, 228-253).
#include "cc_kernels.h"
#include <cstdlib>

namespace {

struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* b) : base(static_cast<char*>(b)) {}
    template <typename T>
    T* take(size_t count) {
        T* ptr = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += cc_align_up(count * sizeof(T), 256);
        return ptr;
    }
};

struct VitWs {
    _Float16* im2col;
    float* h;        // residual stream (frame-major)
    float* h2;       // cluster output (ping-pong)
    _Float16* h16;   // fp16 copy of the residual stream (GEMM A operand; LayerNorm is folded into the GEMM)
    float* st0;      // partial (sum, sumsq) of the rows entering in_proj  [M][CC_LN_MAX_SLOTS][2]
    float* st1;      // ... entering c_fc
    float* sh0;      // per-row constant the fp16 copy entering in_proj was centred on  [M]
    float* sh1;      // ... entering c_fc
    _Float16* qkv;
    _Float16* att;
    _Float16* u;
    void* cluster;
    size_t cluster_bytes;
    size_t total;
};

VitWs carve_vit(const cc_vit_model* m, int B, int T, void* ws) {
    VitWs v{};
    Carver c(ws);
    const int g = m->resolution / m->patch, n = g * g, L0 = n + 1, W = m->width;
    const size_t F = (size_t)B * T, M0 = F * L0;
    v.im2col = c.take<_Float16>(F * n * (m->conv2_weight_f16 ? 9 : 3) * m->patch * m->patch);
    v.h = c.take<float>(M0 * W);
    v.h2 = c.take<float>(M0 * W);
    v.h16 = c.take<_Float16>(M0 * W);
    v.st0 = c.take<float>(M0 * CC_LN_MAX_SLOTS * 2);
    v.st1 = c.take<float>(M0 * CC_LN_MAX_SLOTS * 2);
    v.sh0 = c.take<float>(M0);
    v.sh1 = c.take<float>(M0);
    v.qkv = c.take<_Float16>(M0 * 3 * W);
    v.att = c.take<_Float16>(M0 * W);
    v.u = c.take<_Float16>(M0 * 4 * W);
    // cluster scratch: worst case over the plan
    size_t cb = 0;
    int frames = T, tokens = n;
    for (int i = 0; i < m->layers; ++i) {
        if (m->cluster_tokens[i] > 0) {
            const int Tn = m->cluster_frames[i];
            if (Tn > 0 && frames % Tn == 0) {
                size_t need = cc_cluster_workspace_bytes(B * Tn, (frames / Tn) * tokens, W, m->cluster_pre_norm);
                if (m->cluster_variants && m->cluster_variants[i].algorithm == CC_CLUSTER_SPECTRAL)
                    need += cc_spectral_workspace_bytes(B * Tn, (frames / Tn) * tokens, m->cluster_tokens[i]);
                cb = need > cb ? need : cb;
                frames = Tn;
                tokens = m->cluster_tokens[i];
            }
        }
    }
    v.cluster_bytes = cb;
    v.cluster = c.take<char>(cb);
    v.total = c.off;
    return v;
}

struct BlockCtx {          // one tower's activations for the current block
    float* h;
    _Float16* h16;
    float* st0;
    float* st1;
    float* sh0;
    float* sh1;
    _Float16* qkv;
    _Float16* att;
    _Float16* u;
    int nseq, L, W, heads, causal;
    int slots0, slots1;    // partial-sum slots per row currently held in st0 / st1
    // compacted captions (text tower): device-side row count and per-caption (offset, length); null = dense [nseq, L]
    const int* m_dev;
    const int* seq_off;
    const int* seq_len;
    // Last block of a tower when nobody asked for the hidden state: only sel_rows rows feed the projection head (row
    // m -> sel_map ? sel_map[m] : m * sel_step), so everything after the attention runs on those rows in place
    int sel_rows, sel_step;
    const int* sel_map;
};

// One ResidualAttentionBlock for up to two towers at once (modules/clip.py:240,251).  Every phase is
// ONE launch covering both problems: the text tower (M = 16*32 rows, launch-latency bound on its own:
// 86 kernels of a few microseconds) rides inside the visual tower's launches and fills their tail round.
// LayerNorm never runs as a pass of its own: ln_1 / ln_2 are folded into in_proj / c_fc (the row statistics
// come out of the preceding residual epilogue), which removes two full reads of the fp32 residual stream
// and two launches per block.
int run_block_pair(const cc_block_weights* w0, BlockCtx* c0, const cc_block_weights* w1, BlockCtx* c1,
                   hipStream_t st) {
    if (!w0) { w0 = w1; c0 = c1; w1 = nullptr; c1 = nullptr; }
    const int M0 = c0->nseq * c0->L, M1 = c1 ? c1->nseq * c1->L : 0;
    const int Wa = c0->W, Wb = c1 ? c1->W : 0;
    int rc;
    int slots[2];
    // experiment knob (development builds, -DCC_DEV_KNOBS): CC_UNPAIR bit mask (1 in_proj, 2 out_proj, 4 c_fc, 8 c_proj) - the clustered visual blocks launch
    // those phases separately for the two towers
#ifdef CC_DEV_KNOBS
    static const int unpair_mask = [] { const char* e = getenv("CC_UNPAIR"); return e ? atoi(e) : 0; }();
#else
    constexpr int unpair_mask = 0;
#endif
    const int unpair = (c1 && M0 < 5000) ? unpair_mask : 0;
    auto dispatch = [&](GemmArgs& g0, GemmArgs& g1, int epi, int bit, int* sl) {
        if (!(unpair & bit)) return cc_gemm_dispatch2(g0, c1 ? &g1 : nullptr, epi, 0, st, sl);
        int s0[2] = {0, 0}, s1[2] = {0, 0};
        int r = cc_gemm_dispatch2(g0, nullptr, epi, 0, st, sl ? s0 : nullptr);
        if (r) return r;
        r = cc_gemm_dispatch2(g1, nullptr, epi, 0, st, sl ? s1 : nullptr);
        if (sl) { sl[0] = s0[0]; sl[1] = s1[0]; }
        return r;
    };
    auto base = [&](const BlockCtx* c, int M, const _Float16* A, const void* Wt, const float* bias, void* C, int N, int K) {
        GemmArgs g{};
        g.A = A; g.W = static_cast<const _Float16*>(Wt); g.bias = bias; g.C = C;
        g.M = M; g.N = N; g.K = K; g.ldc = N;
        g.ln_eps = 1e-5f;
        g.m_dev = c->m_dev;
        return g;
    };
    // the phases behind the attention: every row, or the rows the head will read (GemmArgs::row_step / row_map) with the
    // few-rows kernel; a tower whose partner is not in the same mode gets a launch of its own
    auto tail = [&](const BlockCtx* c, int M, const _Float16* A, const void* Wt, const float* bias, void* C, int N, int K) {
        GemmArgs g = base(c, M, A, Wt, bias, C, N, K);
        if (c->sel_rows > 0) { g.M = c->sel_rows; g.m_dev = nullptr; g.row_step = c->sel_step; g.row_map = c->sel_map; }
        return g;
    };
    auto tail_dispatch = [&](GemmArgs& g0, GemmArgs& g1, int epi, int bit, int* sl) {
        const bool s0 = c0->sel_rows > 0, s1 = c1 && c1->sel_rows > 0;
        if (!s0 && !s1) return dispatch(g0, g1, epi, bit, sl);
        if (s0 && (s1 || !c1)) return cc_gemm_rows_dispatch2(g0, c1 ? &g1 : nullptr, epi, st, sl);
        int a[2] = {0, 0}, b[2] = {0, 0};
        int* pa = sl ? a : nullptr;
        int* pb = sl ? b : nullptr;
        int r = s0 ? cc_gemm_rows_dispatch2(g0, nullptr, epi, st, pa) : cc_gemm_dispatch2(g0, nullptr, epi, 0, st, pa);
        if (r) return r;
        r = s1 ? cc_gemm_rows_dispatch2(g1, nullptr, epi, st, pb) : cc_gemm_dispatch2(g1, nullptr, epi, 0, st, pb);
        if (sl) { sl[0] = a[0]; sl[1] = b[0]; }
        return r;
    };
    // ---- q,k,v = in_proj(ln_1(x))   [LayerNorm folded]  and  attn = softmax(q k^T / 8) v
    // One launch where the sequences fit a row tile (L <= 56: the ViT-B/32 frames, every clustered block, the captions): a
    // workgroup owns whole sequences x one head and keeps their q, k, v in LDS (EPI_ATTN_LN); otherwise in_proj writes qkv
    // and the attention kernel reads it back.
    {
        GemmArgs g0 = base(c0, M0, c0->h16, w0->in_proj_ln_weight_f16, w0->in_proj_ln_c2, c0->qkv, 3 * Wa, Wa);
        g0.ln_stats = c0->st0; g0.ln_slots = c0->slots0; g0.ln_c1 = w0->in_proj_ln_c1;
        GemmArgs g1{};
        if (c1) {
            g1 = base(c1, M1, c1->h16, w1->in_proj_ln_weight_f16, w1->in_proj_ln_c2, c1->qkv, 3 * Wb, Wb);
            g1.ln_stats = c1->st0; g1.ln_slots = c1->slots0; g1.ln_c1 = w1->in_proj_ln_c1;
        }
        auto fused = [](GemmArgs g, const BlockCtx* c) {
            g.C = c->att; g.ldc = c->W;
            g.att_L = c->L; g.att_nseq = c->nseq; g.att_causal = c->causal; g.att_seq_off = c->seq_off; g.att_seq_len = c->seq_len;
            return g;
        };
        GemmArgs f0 = fused(g0, c0), f1{};
        if (c1) f1 = fused(g1, c1);
        if (!(unpair & 1) && c0->heads * 64 == Wa && (!c1 || c1->heads * 64 == Wb) && cc_gemm_attn_applies(f0, c1 ? &f1 : nullptr)) {
            rc = cc_gemm_attn_dispatch2(f0, c1 ? &f1 : nullptr, st);
            if (rc) return rc;
        } else {
            rc = dispatch(g0, g1, EPI_F16_LN, 1, nullptr);
            if (rc) return rc;
            AttArgs a0{c0->qkv, c0->att, c0->nseq, c0->L, c0->heads, c0->W, c0->causal, 0, 0, c0->seq_off, c0->seq_len};
            AttArgs a1{};
            if (c1) a1 = AttArgs{c1->qkv, c1->att, c1->nseq, c1->L, c1->heads, c1->W, c1->causal, 0, 0, c1->seq_off, c1->seq_len};
            rc = cc_launch_attention2(a0, c1 ? &a1 : nullptr, st);
            if (rc) return rc;
        }
    }
    // ---- x = x + out_proj(attn)   [+ fp16 copy and row statistics for ln_2]
    {
        GemmArgs g0 = tail(c0, M0, c0->att, w0->out_proj_weight_f16, w0->out_proj_bias, c0->h, Wa, Wa);
        g0.c16 = c0->h16; g0.stats_out = c0->st1;
        g0.shift_in = c0->sh0; g0.shift_stats = c0->st0; g0.shift_slots = c0->slots0; g0.shift_out = c0->sh1;
        GemmArgs g1{};
        if (c1) {
            g1 = tail(c1, M1, c1->att, w1->out_proj_weight_f16, w1->out_proj_bias, c1->h, Wb, Wb);
            g1.c16 = c1->h16; g1.stats_out = c1->st1;
            g1.shift_in = c1->sh0; g1.shift_stats = c1->st0; g1.shift_slots = c1->slots0; g1.shift_out = c1->sh1;
        }
        rc = tail_dispatch(g0, g1, EPI_F32_RESID_STATS, 2, slots);
        if (rc) return rc;
        c0->slots1 = slots[0];
        if (c1) c1->slots1 = slots[1];
    }
    // ---- u = QuickGELU(c_fc(ln_2(x)))   [LayerNorm folded]
    {
        GemmArgs g0 = tail(c0, M0, c0->h16, w0->c_fc_ln_weight_f16, w0->c_fc_ln_c2, c0->u, 4 * Wa, Wa);
        g0.ln_stats = c0->st1; g0.ln_slots = c0->slots1; g0.ln_c1 = w0->c_fc_ln_c1;
        GemmArgs g1{};
        if (c1) {
            g1 = tail(c1, M1, c1->h16, w1->c_fc_ln_weight_f16, w1->c_fc_ln_c2, c1->u, 4 * Wb, Wb);
            g1.ln_stats = c1->st1; g1.ln_slots = c1->slots1; g1.ln_c1 = w1->c_fc_ln_c1;
        }
        rc = tail_dispatch(g0, g1, EPI_F16_GELU_LN, 4, nullptr);
        if (rc) return rc;
    }
    // ---- x = x + c_proj(u)   [+ fp16 copy and row statistics for the next block's ln_1]
    {
        GemmArgs g0 = tail(c0, M0, c0->u, w0->c_proj_weight_f16, w0->c_proj_bias, c0->h, Wa, 4 * Wa);
        g0.c16 = c0->h16; g0.stats_out = c0->st0;
        g0.shift_in = c0->sh1; g0.shift_stats = c0->st1; g0.shift_slots = c0->slots1; g0.shift_out = c0->sh0;
        GemmArgs g1{};
        if (c1) {
            g1 = tail(c1, M1, c1->u, w1->c_proj_weight_f16, w1->c_proj_bias, c1->h, Wb, 4 * Wb);
            g1.c16 = c1->h16; g1.stats_out = c1->st0;
            g1.shift_in = c1->sh1; g1.shift_stats = c1->st1; g1.shift_slots = c1->slots1; g1.shift_out = c1->sh0;
        }
        rc = tail_dispatch(g0, g1, EPI_F32_RESID_STATS, 8, slots);
        if (rc) return rc;
        c0->slots0 = slots[0];
        if (c1) c1->slots0 = slots[1];
    }
    return CC_OK;
}

struct TextWs {
    float* h;
    _Float16* h16;
    float* st0;
    float* st1;
    float* sh0;
    float* sh1;
    _Float16* qkv;
    _Float16* att;
    _Float16* u;
    int* eot;
    int* seq_off;    // compacted captions: first row of caption b
    int* seq_len;    //                      its length (EOT position + 1)
    int* mcount;     //                      total number of kept rows
    size_t total;
};

TextWs carve_text(const cc_text_model* m, int Bt, int Lt, void* ws) {
    TextWs t{};
    Carver c(ws);
    const size_t M = (size_t)Bt * Lt, W = m->width;
    t.h = c.take<float>(M * W);
    t.h16 = c.take<_Float16>(M * W);
    t.st0 = c.take<float>(M * CC_LN_MAX_SLOTS * 2);
    t.st1 = c.take<float>(M * CC_LN_MAX_SLOTS * 2);
    t.sh0 = c.take<float>(M);
    t.sh1 = c.take<float>(M);
    t.qkv = c.take<_Float16>(M * 3 * W);
    t.att = c.take<_Float16>(M * W);
    t.u = c.take<_Float16>(M * 4 * W);
    t.eot = c.take<int>(Bt);
    t.seq_off = c.take<int>(Bt);
    t.seq_len = c.take<int>(Bt);
    t.mcount = c.take<int>(1);
    t.total = c.off;
    return t;
}

bool vit_ok(const cc_vit_model* m) {
    if (m->layers <= 0 || m->layers > CC_MAX_LAYERS || m->width != m->heads * 64) return false;
    return !(m->resolution % m->patch || (m->patch & 7) || (m->width % 64) || ((3 * m->patch * m->patch) % 64));
}

bool text_ok(const cc_text_model* m, int Lt) {
    return m->layers > 0 && m->layers <= CC_MAX_LAYERS && m->width == m->heads * 64 && Lt <= m->context_length;
}

// Both towers, block i of the one paired with block i of the other (either may be absent).
int encode_towers(const cc_vit_model* vm, const cc_frames* video, int B, int T, float* vfeat, float* hidden_out,
                  int64_t* medoids_out, const int64_t* forced_medoids, const cc_text_model* tm, const int64_t* ids,
                  int Bt, int Lt, float* tfeat, float* text_hidden_out, void* ws, size_t ws_bytes, hipStream_t st) {
    VitWs v{};
    TextWs t{};
    size_t off = 0;
    if (vm) { v = carve_vit(vm, B, T, ws); off = v.total; }
    if (tm) { t = carve_text(tm, Bt, Lt, static_cast<char*>(ws) + off); off += t.total; }
    if (!ws || ws_bytes < off) return CC_ERR_WORKSPACE;
    int rc = CC_OK;
    float* h = nullptr;
    float* hother = nullptr;
    int frames = T, tokens = 0, W = 0;
    const int vl = vm ? vm->layers : 0, tl = tm ? tm->layers : 0;
    // Caption compaction (see TextEmbedArgs): the text tower runs on the rows up to each caption's EOT only - the launches
    // are sized for Bt * Lt rows, the kernels read the real count from the device, so nothing synchronises and a
    // captured graph stays valid for any batch.  Off when the caller wants the full hidden state.
    const bool compact = tm && !text_hidden_out && Bt <= 256 && !(tm->row_policy & CC_ROWS_ALL_TEXT);
    BlockCtx cv{}, ct{};
    cv.slots0 = ct.slots0 = 1;
    if (tm) {
        ct.h = t.h; ct.h16 = t.h16; ct.st0 = t.st0; ct.st1 = t.st1; ct.sh0 = t.sh0; ct.sh1 = t.sh1;
        ct.qkv = t.qkv; ct.att = t.att; ct.u = t.u;
        ct.nseq = Bt; ct.L = Lt; ct.W = tm->width; ct.heads = tm->heads; ct.causal = 1;
        if (compact) { ct.m_dev = t.mcount; ct.seq_off = t.seq_off; ct.seq_len = t.seq_len; }
    }
    if (vm) {
        const int g = vm->resolution / vm->patch, n = g * g, F = B * T;
        W = vm->width;
        tokens = n;
        // patch embedding: conv1 as im2col GEMM, + positional embedding, CLS row, ln_pre (clip.py:324-338)
        const bool patch3d = vm->conv2_weight_f16 != nullptr;      // linear_patch '3d' (clip.py:306-317)
        rc = patch3d ? cc_launch_im2col3d(*video, v.im2col, F, T, vm->resolution, vm->patch, st)
                     : cc_launch_im2col(*video, v.im2col, F, vm->resolution, vm->patch, st);
        if (rc) return rc;
        GemmArgs ga{};
        ga.A = v.im2col;
        ga.W = static_cast<const _Float16*>(patch3d ? vm->conv2_weight_f16 : vm->conv1_weight_f16);
        ga.C = v.h;
        ga.pos = vm->positional_embedding;
        ga.M = F * n; ga.N = W; ga.K = (patch3d ? 9 : 3) * vm->patch * vm->patch; ga.ldc = W;
        ga.patch_n = n;
        rc = cc_gemm_dispatch(ga, EPI_F32_PATCH, 0, st);
        if (rc) return rc;
        h = v.h;
        hother = v.h2;
    }
    {   // ln_pre in place (fp32; the CLS rows = class_embedding + positional_embedding[0] are formed inside) + fp16 copy +
        // row statistics for block 1's folded ln_1, and the text embedding with the same by-products - one launch
        LnArgs a{};
        TextEmbedArgs te{};
        if (vm) {
            const int n = tokens, F = B * T;
            a = LnArgs{v.h, W, vm->ln_pre_weight, vm->ln_pre_bias, v.h, W, F * (n + 1), W, v.h16, v.st0, v.sh0,
                       vm->class_embedding, vm->positional_embedding, n + 1};
        }
        if (tm)
            te = TextEmbedArgs{reinterpret_cast<const long long*>(ids), tm->token_embedding, tm->positional_embedding,
                               t.h, t.eot, Bt, Lt, tm->width, t.h16, t.st0, t.sh0,
                               compact ? t.seq_off : nullptr, compact ? t.seq_len : nullptr, compact ? t.mcount : nullptr,
                               tm->vocab_size};
        rc = (vm && tm) ? cc_launch_pre_stage(a, te, 1e-5f, st)
                        : vm ? cc_launch_layernorm2(a, nullptr, 1e-5f, 0, st) : cc_launch_text_embed(te, st);
        if (rc) return rc;
    }
    // medoids_out receives the ids of the LAST k-medoids block only (it is sized for that block); forced_medoids holds the id
    // tensors of ALL cluster blocks back to back, in block order ([B * T_new_i, K_i] int64 each: round 5 - one block before)
    int last_kmed = -1;
    for (int i = 0; i < vl; ++i)
        if (vm->cluster_tokens[i] > 0) {
            const cc_cluster_variant* var = vm->cluster_variants ? &vm->cluster_variants[i] : nullptr;
            if (!var || var->algorithm == CC_CLUSTER_KMEDOIDS || var->algorithm == CC_CLUSTER_SPECTRAL) last_kmed = i;
        }
    size_t forced_off = 0;      // first id of the current cluster block inside forced_medoids
    int ti = 0;                 // next text block
    for (int i = 0; i < vl || ti < tl; ++i) {
        const bool hv = i < vl;
        if (hv) {
            if (vm->cluster_tokens[i] > 0) {        // token cluster before the attention of this block (clip.py:236-242)
                const int Tn = vm->cluster_frames[i], K = vm->cluster_tokens[i];
                if (Tn <= 0 || frames % Tn) return CC_ERR_INVALID;
                const cc_cluster_variant* var = vm->cluster_variants ? &vm->cluster_variants[i] : nullptr;
                if (var && var->algorithm == CC_CLUSTER_POOLING && K != tokens) return CC_ERR_INVALID;
                // (the gather / aggregation launch also writes the fp16 copy + row statistics of the new rows)
                cc_cluster_variant dflt{};
                dflt.algorithm = CC_CLUSTER_KMEDOIDS;
                dflt.aggregation = CC_AGGREGATE_MEDOID;
                if (forced_medoids && (!var || (var->algorithm == CC_CLUSTER_KMEDOIDS &&
                                                var->aggregation == CC_AGGREGATE_MEDOID && !var->cluster_embed &&
                                                !var->cls_multiplier)))
                    rc = cc_token_gather_rows(h, W, (int64_t)(tokens + 1) * W, B, frames, Tn, tokens, W, K,
                                              forced_medoids + forced_off, hother, W, (int64_t)(K + 1) * W, v.h16, v.st0, v.sh0, st);
                else if (forced_medoids)
                    rc = CC_ERR_UNSUPPORTED;
                else
                    rc = cc_token_cluster_variant_rows(h, W, (int64_t)(tokens + 1) * W, B, frames, Tn, tokens, W, K,
                                                       vm->cluster_metric, vm->cluster_norm_p, vm->cluster_threshold,
                                                       vm->cluster_iter_limit, vm->cluster_split_size,
                                                       vm->cluster_pre_norm, var ? var : &dflt, hother, W,
                                                       (int64_t)(K + 1) * W, i == last_kmed ? medoids_out : nullptr,
                                                       nullptr, nullptr, v.cluster, v.cluster_bytes, v.h16, v.st0,
                                                       v.sh0, st);
                if (rc) return rc;
                if (var && var->mean_residual) {
                    // clip.py:239-242: x = res_x + attention(ln_1(x')) - ln_1 reads the clustered rows (their fp16 copy and
                    // statistics are written), the fp32 residual stream restarts from the frame means of every token
                    if (K != tokens) return CC_ERR_INVALID;                 // cluster.py:229
                    cc_cluster_variant pool{};
                    pool.algorithm = CC_CLUSTER_POOLING;
                    rc = cc_token_cluster_variant_rows(h, W, (int64_t)(tokens + 1) * W, B, frames, Tn, tokens, W, K,
                                                       vm->cluster_metric, vm->cluster_norm_p, vm->cluster_threshold,
                                                       vm->cluster_iter_limit, vm->cluster_split_size, vm->cluster_pre_norm,
                                                       &pool, hother, W, (int64_t)(K + 1) * W, nullptr, nullptr, nullptr,
                                                       v.cluster, v.cluster_bytes, nullptr, nullptr, nullptr, st);
                    if (rc) return rc;
                }
                forced_off += (size_t)B * Tn * K;
                float* tmp = h; h = hother; hother = tmp;
                frames = Tn;
                tokens = K;
                cv.slots0 = 1;
            }
            cv.h = h; cv.h16 = v.h16; cv.st0 = v.st0; cv.st1 = v.st1; cv.sh0 = v.sh0; cv.sh1 = v.sh1;
            cv.qkv = v.qkv; cv.att = v.att; cv.u = v.u;
            cv.nseq = B * frames; cv.L = tokens + 1; cv.W = W; cv.heads = vm->heads; cv.causal = 0;
        }
        const bool ht = ti < tl;
        // last block, nobody wants the hidden state: everything behind the attention on the CLS / EOT rows only
        auto few_rows_ok = [](int Wd) {
            return cc_gemm_rows_ok(Wd, Wd, EPI_F32_RESID_STATS) && cc_gemm_rows_ok(4 * Wd, Wd, EPI_F16_GELU_LN) &&
                   cc_gemm_rows_ok(Wd, 4 * Wd, EPI_F32_RESID_STATS);
        };
        if (hv && i == vl - 1 && !hidden_out && !(vm->row_policy & CC_ROWS_ALL_LAST_BLOCK) && cv.L > 1 && few_rows_ok(W)) {
            cv.sel_rows = cv.nseq;
            cv.sel_step = cv.L;
        }
        if (ht && ti == tl - 1 && compact && !(tm->row_policy & CC_ROWS_ALL_LAST_BLOCK) && few_rows_ok(tm->width)) {
            ct.sel_rows = Bt;
            ct.sel_map = t.eot;
        }
        rc = run_block_pair(hv ? &vm->blocks[i] : nullptr, hv ? &cv : nullptr, ht ? &tm->blocks[ti] : nullptr,
                            ht ? &ct : nullptr, st);
        if (rc) return rc;
        if (ht) ++ti;
    }
    // ln_post + proj on the CLS rows only (clip.py:463-464); ln_final + text_projection on the EOT rows only
    // (clip.py:480-484) - one launch for both heads
    HeadArgs hv{}, ht{};
    if (vm) hv = HeadArgs{h, tokens + 1, nullptr, vm->ln_post_weight, vm->ln_post_bias, vm->proj, vfeat, B * frames, W, vm->embed_dim};
    // (compacted captions: eot[b] is already the absolute row of the EOT token)
    if (tm) ht = HeadArgs{t.h, compact ? 0 : Lt, t.eot, tm->ln_final_weight, tm->ln_final_bias, tm->text_projection, tfeat, Bt, tm->width, tm->embed_dim};
    rc = cc_launch_head_project2(vm ? hv : ht, (vm && tm) ? &ht : nullptr, st);
    if (rc) return rc;
    if (vm && hidden_out && hipMemcpyAsync(hidden_out, h, (size_t)B * frames * (tokens + 1) * W * sizeof(float),
                                           hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CC_ERR_HIP;
    if (tm && text_hidden_out && hipMemcpyAsync(text_hidden_out, t.h, (size_t)Bt * Lt * tm->width * sizeof(float),
                                                hipMemcpyDeviceToDevice, st) != hipSuccess)
        return CC_ERR_HIP;
    return CC_OK;
}

}  // namespace

extern "C" {

size_t cc_vit_workspace_bytes(const cc_vit_model* m, int32_t B, int32_t T) {
    if (!m || B <= 0 || T <= 0 || m->patch <= 0 || m->resolution % m->patch) return 0;
    return carve_vit(m, B, T, nullptr).total;
}

int64_t cc_vit_forced_medoids_count(const cc_vit_model* m, int32_t B) {
    if (!m || B <= 0 || m->layers < 0 || m->layers > CC_MAX_LAYERS) return CC_ERR_INVALID;
    int64_t n = 0;
    for (int i = 0; i < m->layers; ++i)
        if (m->cluster_tokens[i] > 0) n += (int64_t)B * m->cluster_frames[i] * m->cluster_tokens[i];
    return n;
}

int cc_vit_encode_frames(const cc_vit_model* m, const cc_frames* frames, int32_t B, int32_t T, float* features,
                         float* hidden_out, int64_t* medoids_out, const int64_t* forced_medoids, void* ws,
                         size_t ws_bytes, void* stream) {
    if (!m || !frames || !frames->data || !features || !m->blocks || B <= 0 || T <= 0) return CC_ERR_INVALID;
    if (!vit_ok(m)) return CC_ERR_UNSUPPORTED;
    return encode_towers(m, frames, B, T, features, hidden_out, medoids_out, forced_medoids, nullptr, nullptr, 0, 0,
                         nullptr, nullptr, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int cc_vit_encode(const cc_vit_model* m, const float* video, int32_t B, int32_t T, float* features,
                  float* hidden_out, int64_t* medoids_out, const int64_t* forced_medoids, void* ws, size_t ws_bytes,
                  void* stream) {
    cc_frames fr{};
    fr.data = video;
    fr.format = CC_FRAMES_F32_CHW;
    return cc_vit_encode_frames(m, &fr, B, T, features, hidden_out, medoids_out, forced_medoids, ws, ws_bytes, stream);
}

size_t cc_text_workspace_bytes(const cc_text_model* m, int32_t Bt, int32_t Lt) {
    if (!m || Bt <= 0 || Lt <= 0) return 0;
    return carve_text(m, Bt, Lt, nullptr).total;
}

int cc_text_encode_hidden(const cc_text_model* m, const int64_t* ids, int32_t Bt, int32_t Lt, float* features,
                          float* hidden_out, void* ws, size_t ws_bytes, void* stream) {
    if (!m || !ids || !features || !m->blocks || Bt <= 0 || Lt <= 0) return CC_ERR_INVALID;
    if (Lt > m->context_length) return CC_ERR_INVALID;
    if (!text_ok(m, Lt)) return CC_ERR_UNSUPPORTED;
    return encode_towers(nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, m, ids, Bt, Lt, features, hidden_out,
                         ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int cc_text_encode(const cc_text_model* m, const int64_t* ids, int32_t Bt, int32_t Lt, float* features, void* ws,
                   size_t ws_bytes, void* stream) {
    return cc_text_encode_hidden(m, ids, Bt, Lt, features, nullptr, ws, ws_bytes, stream);
}

int cc_head_project_f32(const float* h, int32_t row_mul, const int32_t* row_idx, const float* gamma, const float* beta,
                        const float* proj, float* out, int32_t R, int32_t W, int32_t E, void* stream) {
    if (!h || !gamma || !beta || !proj || !out || R <= 0 || W <= 0 || E <= 0 || row_mul <= 0) return CC_ERR_INVALID;
    return cc_launch_head_project(h, row_mul, row_idx, gamma, beta, proj, out, R, W, E, static_cast<hipStream_t>(stream));
}

size_t cc_clip_workspace_bytes(const cc_vit_model* vm, int32_t B, int32_t T, const cc_text_model* tm, int32_t Bt,
                               int32_t Lt) {
    return cc_vit_workspace_bytes(vm, B, T) + cc_text_workspace_bytes(tm, Bt, Lt);
}

int cc_clip_encode_frames(const cc_vit_model* vm, const cc_frames* frames, int32_t B, int32_t T,
                          float* visual_features, int64_t* medoids_out, const int64_t* forced_medoids,
                          const cc_text_model* tm, const int64_t* ids, int32_t Bt, int32_t Lt, float* text_features,
                          void* ws, size_t ws_bytes, void* stream) {
    if (!vm || !tm || !frames || !frames->data || !ids || !visual_features || !text_features || !vm->blocks ||
        !tm->blocks)
        return CC_ERR_INVALID;
    if (B <= 0 || T <= 0 || Bt <= 0 || Lt <= 0 || Lt > tm->context_length) return CC_ERR_INVALID;
    if (!vit_ok(vm) || !text_ok(tm, Lt)) return CC_ERR_UNSUPPORTED;
    return encode_towers(vm, frames, B, T, visual_features, nullptr, medoids_out, forced_medoids, tm, ids, Bt, Lt,
                         text_features, nullptr, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

int cc_clip_encode(const cc_vit_model* vm, const float* video, int32_t B, int32_t T, float* visual_features,
                   int64_t* medoids_out, const cc_text_model* tm, const int64_t* ids, int32_t Bt, int32_t Lt,
                   float* text_features, void* ws, size_t ws_bytes, void* stream) {
    cc_frames fr{};
    fr.data = video;
    fr.format = CC_FRAMES_F32_CHW;
    return cc_clip_encode_frames(vm, &fr, B, T, visual_features, medoids_out, nullptr, tm, ids, Bt, Lt, text_features,
                                 ws, ws_bytes, stream);
}

}  // extern "C"
