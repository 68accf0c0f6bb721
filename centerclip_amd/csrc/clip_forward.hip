// Host-side orchestration of the CLIP visual / text forward: one C call enqueues the whole
// encoder on a HIP stream (no host synchronisation, no allocation - the workspace is caller
// owned), so it can be captured into a hipGraph as is.
//
// Activations are frame-major ([frame, token, width], "NLD"): attention works on contiguous
// per-frame rows and the token-cluster op reads/writes the same buffer through strides.
// Reference call stack: CLIP.encode_image -> VisualTransformer.forward -> Transformer ->
// ResidualAttentionBlock.forward (modules/clip.py:460-469, 304-349, 256-269, 228-253).
#include "cc_kernels.h"

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
    _Float16* xn;
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
    v.im2col = c.take<_Float16>(F * n * 3 * m->patch * m->patch);
    v.h = c.take<float>(M0 * W);
    v.h2 = c.take<float>(M0 * W);
    v.xn = c.take<_Float16>(M0 * W);
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
                const size_t need = cc_cluster_workspace_bytes(B * Tn, (frames / Tn) * tokens, W, m->cluster_pre_norm);
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

int run_block(const cc_block_weights& w, float* h, _Float16* xn, _Float16* qkv, _Float16* att, _Float16* u, int nseq,
              int L, int W, int heads, int causal, hipStream_t st) {
    const int M = nseq * L;
    int rc;
    // x = x + attn(ln_1(x))                                        modules/clip.py:240
    rc = cc_layernorm_f32(h, W, w.ln_1_weight, w.ln_1_bias, xn, W, M, W, 1e-5f, 1, st);
    if (rc) return rc;
    rc = cc_linear_f16(xn, w.in_proj_weight_f16, w.in_proj_bias, qkv, M, 3 * W, W, 3 * W, EPI_F16, 0, st);
    if (rc) return rc;
    rc = cc_attention_f16(qkv, att, nseq, L, heads, W, causal, st);
    if (rc) return rc;
    rc = cc_linear_f16(att, w.out_proj_weight_f16, w.out_proj_bias, h, M, W, W, W, EPI_F32_RESID, 0, st);
    if (rc) return rc;
    // x = x + c_proj(QuickGELU(c_fc(ln_2(x))))                      modules/clip.py:251
    rc = cc_layernorm_f32(h, W, w.ln_2_weight, w.ln_2_bias, xn, W, M, W, 1e-5f, 1, st);
    if (rc) return rc;
    rc = cc_linear_f16(xn, w.c_fc_weight_f16, w.c_fc_bias, u, M, 4 * W, W, 4 * W, EPI_F16_GELU, 0, st);
    if (rc) return rc;
    return cc_linear_f16(u, w.c_proj_weight_f16, w.c_proj_bias, h, M, W, 4 * W, W, EPI_F32_RESID, 0, st);
}

}  // namespace

extern "C" {

size_t cc_vit_workspace_bytes(const cc_vit_model* m, int32_t B, int32_t T) {
    if (!m || B <= 0 || T <= 0 || m->patch <= 0 || m->resolution % m->patch) return 0;
    return carve_vit(m, B, T, nullptr).total;
}

int cc_vit_encode(const cc_vit_model* m, const float* video, int32_t B, int32_t T, float* features,
                  float* hidden_out, int64_t* medoids_out, const int64_t* forced_medoids, void* ws, size_t ws_bytes,
                  void* stream) {
    if (!m || !video || !features || !m->blocks || B <= 0 || T <= 0) return CC_ERR_INVALID;
    if (m->layers <= 0 || m->layers > CC_MAX_LAYERS || m->width != m->heads * 64) return CC_ERR_UNSUPPORTED;
    if (m->resolution % m->patch || (m->patch & 7) || (m->width % 64) || ((3 * m->patch * m->patch) % 64))
        return CC_ERR_UNSUPPORTED;
    VitWs v = carve_vit(m, B, T, ws);
    if (!ws || ws_bytes < v.total) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int g = m->resolution / m->patch, n = g * g, W = m->width, F = B * T;
    int rc;

    // ---- patch embedding: conv1 as im2col GEMM, + positional embedding, CLS row, ln_pre (clip.py:324-338)
    rc = cc_launch_im2col(video, v.im2col, F, m->resolution, m->patch, st);
    if (rc) return rc;
    {
        GemmArgs ga{};
        ga.A = v.im2col;
        ga.W = static_cast<const _Float16*>(m->conv1_weight_f16);
        ga.bias = nullptr;
        ga.C = v.h;
        ga.pos = m->positional_embedding;
        ga.M = F * n; ga.N = W; ga.K = 3 * m->patch * m->patch; ga.ldc = W;
        ga.patch_n = n;
        rc = cc_gemm_dispatch(ga, EPI_F32_PATCH, 0, st);
        if (rc) return rc;
    }
    rc = cc_launch_cls_pos(v.h, m->class_embedding, m->positional_embedding, F, n + 1, W, st);
    if (rc) return rc;
    rc = cc_layernorm_f32(v.h, W, m->ln_pre_weight, m->ln_pre_bias, v.h, W, F * (n + 1), W, 1e-5f, 0, st);
    if (rc) return rc;

    // ---- transformer blocks, token cluster before the attention of the planned blocks (clip.py:236-242)
    float* h = v.h;
    float* hother = v.h2;
    int frames = T, tokens = n;
    for (int i = 0; i < m->layers; ++i) {
        if (m->cluster_tokens[i] > 0) {
            const int Tn = m->cluster_frames[i], K = m->cluster_tokens[i];
            if (Tn <= 0 || frames % Tn) return CC_ERR_INVALID;
            if (forced_medoids)
                rc = cc_token_gather_f32(h, W, (int64_t)(tokens + 1) * W, B, frames, Tn, tokens, W, K, forced_medoids,
                                         hother, W, (int64_t)(K + 1) * W, st);
            else
                rc = cc_token_cluster_f32(h, W, (int64_t)(tokens + 1) * W, B, frames, Tn, tokens, W, K,
                                          m->cluster_metric, m->cluster_norm_p, m->cluster_threshold,
                                          m->cluster_iter_limit, m->cluster_split_size, m->cluster_pre_norm, hother, W,
                                          (int64_t)(K + 1) * W, medoids_out, nullptr, nullptr, v.cluster,
                                          v.cluster_bytes, st);
            if (rc) return rc;
            float* t = h; h = hother; hother = t;
            frames = Tn;
            tokens = K;
        }
        rc = run_block(m->blocks[i], h, v.xn, v.qkv, v.att, v.u, B * frames, tokens + 1, W, m->heads, 0, st);
        if (rc) return rc;
    }
    // ---- ln_post + proj on the CLS rows only (clip.py:463-464)
    rc = cc_launch_head_project(h, tokens + 1, nullptr, m->ln_post_weight, m->ln_post_bias, m->proj, features,
                                B * frames, W, m->embed_dim, st);
    if (rc) return rc;
    if (hidden_out) {
        if (hipMemcpyAsync(hidden_out, h, (size_t)B * frames * (tokens + 1) * W * sizeof(float),
                           hipMemcpyDeviceToDevice, st) != hipSuccess)
            return CC_ERR_HIP;
    }
    return CC_OK;
}

size_t cc_text_workspace_bytes(const cc_text_model* m, int32_t Bt, int32_t Lt) {
    if (!m || Bt <= 0 || Lt <= 0) return 0;
    Carver c(nullptr);
    const size_t M = (size_t)Bt * Lt, W = m->width;
    c.take<float>(M * W);
    c.take<_Float16>(M * W);
    c.take<_Float16>(M * 3 * W);
    c.take<_Float16>(M * W);
    c.take<_Float16>(M * 4 * W);
    c.take<int>(Bt);
    return c.off;
}

int cc_text_encode(const cc_text_model* m, const int64_t* ids, int32_t Bt, int32_t Lt, float* features, void* ws,
                   size_t ws_bytes, void* stream) {
    if (!m || !ids || !features || !m->blocks || Bt <= 0 || Lt <= 0) return CC_ERR_INVALID;
    if (Lt > m->context_length) return CC_ERR_INVALID;
    if (m->layers <= 0 || m->layers > CC_MAX_LAYERS || m->width != m->heads * 64) return CC_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < cc_text_workspace_bytes(m, Bt, Lt)) return CC_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Carver c(ws);
    const size_t M = (size_t)Bt * Lt, W = m->width;
    float* h = c.take<float>(M * W);
    _Float16* xn = c.take<_Float16>(M * W);
    _Float16* qkv = c.take<_Float16>(M * 3 * W);
    _Float16* att = c.take<_Float16>(M * W);
    _Float16* u = c.take<_Float16>(M * 4 * W);
    int* eot = c.take<int>(Bt);
    int rc = cc_launch_text_embed(reinterpret_cast<const long long*>(ids), m->token_embedding,
                                  m->positional_embedding, h, eot, Bt, Lt, (int)W, st);
    if (rc) return rc;
    for (int i = 0; i < m->layers; ++i) {
        rc = run_block(m->blocks[i], h, xn, qkv, att, u, Bt, Lt, (int)W, m->heads, 1, st);
        if (rc) return rc;
    }
    return cc_launch_head_project(h, Lt, eot, m->ln_final_weight, m->ln_final_bias, m->text_projection, features, Bt,
                                  (int)W, m->embed_dim, st);
}

}  // extern "C"
