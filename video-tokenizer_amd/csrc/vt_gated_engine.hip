// Host-side engine for a stack of the TiTok-style gated layers: ResidualAttentionBlock.forward of
// /root/reference/models/model_new/base/transformer.py:66-91 (Attn :32-63, ffd :20-29) and its backward as ONE enqueue per
// direction.  Replaces the Python composition of round 1 (titok.py::GatedLayer, ~50 launches per layer and direction issued
// from Python, host-bound at the `small` model size).  Nothing here allocates or synchronises: a sequence of launches of the
// kernels in vt_gemm*.hip, vt_gated.hip, vt_attention.hip, vt_norm.hip on the caller's stream, over one caller-owned workspace.
#include <math.h>

#include <vector>

#include "vt_common.h"

#define TRY(x)                  \
    do {                        \
        int rc__ = (x);         \
        if (rc__) return rc__;  \
    } while (0)
#define WS(T, off) ((T*)((char*)ws + (off)))

static inline size_t up(size_t a, size_t b) { return (a + b - 1) / b * b; }

struct GLayerBufs {
    // bf16 operand copies of the weights: [N, K] for the forward, [K, N] for the input gradients
    size_t qkv_wb, qkv_wt, out_wb, out_wt, fc1_wb, fc1_wt, fc2_wb, fc2_wt;
    // saved activations
    size_t x_in, xb, qkvg, qkv, o, lse, og, x1, y, mean, rstd, h, a;
};

struct vtGatedStack {
    vtGatedStackConfig c;
    int M, ipad;
    std::vector<GLayerBufs> L;
    size_t x_out;                 // fp32 [M, D] output of the last layer
    // backward scratch (one set, reused by every layer)
    size_t d2, gb, da, dh, dy, dx1, dx1b, dog, d_o, dqkv, dqkvg, dwfc2_pad, delta, ln_ws, qk_ws;
    size_t ws_bytes;
};

extern "C" int vt_gated_stack_create(const vtGatedStackConfig* cfg, vtGatedStack** out) {
    VT_CHECK_ARG(cfg && out, "vt_gated_stack_create: null pointer");
    const vtGatedStackConfig c = *cfg;
    VT_CHECK_ARG(c.B > 0 && c.L > 0 && c.depth > 0 && c.H > 0 && c.D == 64 * c.H, "vt_gated_stack_create: B=%d L=%d D=%d H=%d depth=%d (D must be 64 * H)", c.B,
                 c.L, c.D, c.H, c.depth);
    VT_CHECK_ARG(((int64_t)c.B * c.L) % 64 == 0, "vt_gated_stack_create: B * L = %lld must be a multiple of 64", (long long)c.B * c.L);
    VT_CHECK_ARG(c.D == 256 || c.D == 512 || c.D == 768 || c.D == 1024, "vt_gated_stack_create: width %d unsupported (256, 512, 768, 1024)", c.D);
    VT_CHECK_ARG(c.inner > 0 && c.inner % 8 == 0, "vt_gated_stack_create: inner=%d must be a positive multiple of 8", c.inner);
    vtGatedStack* t = new vtGatedStack();
    t->c = c;
    t->M = c.B * c.L;
    t->ipad = (int)up(c.inner, 64);
    const size_t M = t->M, D = c.D, I2 = 2 * (size_t)c.inner, ip = t->ipad;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += up(bytes, 256); return o; };
    t->L.resize(c.depth);
    for (auto& l : t->L) {
        l.qkv_wb = take(4 * D * D * 2); l.qkv_wt = take(D * 4 * D * 2);
        l.out_wb = take(D * D * 2); l.out_wt = take(D * D * 2);
        l.fc1_wb = take(I2 * D * 2); l.fc1_wt = take(D * I2 * 2);
        l.fc2_wb = take(D * ip * 2); l.fc2_wt = take(ip * D * 2);
        l.x_in = take(M * D * 4); l.xb = take(M * D * 2);
        l.qkvg = take(M * 4 * D * 2); l.qkv = take(M * 3 * D * 2);
        l.o = take(M * D * 2); l.lse = take((size_t)c.B * c.H * c.L * 4); l.og = take(M * D * 2);
        l.x1 = take(M * D * 4); l.y = take(M * D * 2); l.mean = take(M * 4); l.rstd = take(M * 4);
        l.h = take(M * I2 * 2); l.a = take(M * ip * 2);
    }
    t->x_out = take(M * D * 4);
    t->d2 = take(M * D * 4); t->gb = take(M * D * 2); t->da = take(M * ip * 2); t->dh = take(M * I2 * 2); t->dy = take(M * D * 2);
    t->dx1 = take(M * D * 4); t->dx1b = take(M * D * 2); t->dog = take(M * D * 2); t->d_o = take(M * D * 2);
    t->dqkv = take(M * 3 * D * 2); t->dqkvg = take(M * 4 * D * 2);
    t->dwfc2_pad = take(D * ip * 4);
    t->delta = take((size_t)c.B * c.H * c.L * 4);
    t->ln_ws = take(vt_layernorm_bwd_workspace_bytes(c.D));
    t->qk_ws = take(vt_qknorm_rope_bwd_workspace_bytes());
    t->ws_bytes = off;
    *out = t;
    return VT_OK;
}

extern "C" void vt_gated_stack_destroy(vtGatedStack* t) { delete t; }
extern "C" size_t vt_gated_stack_workspace_bytes(const vtGatedStack* t) { return t ? t->ws_bytes : 0; }

extern "C" int vt_gated_stack_init_workspace(vtGatedStack* t, void* ws, vtStream stream) {
    VT_CHECK_ARG(t && ws, "vt_gated_stack_init_workspace: null pointer");
    // zero once: the 64-padding columns of `a` / rows of fc2_wt are never written afterwards and must contribute nothing
    if (hipMemsetAsync(ws, 0, t->ws_bytes, (hipStream_t)stream) != hipSuccess) { vt_set_error("vt_gated_stack_init_workspace: memset failed"); return VT_ERR_LAUNCH; }
    return VT_OK;
}

static vtGemmNT gnt(const void* A, int64_t lda, const void* B, int64_t ldb, int M, int N, int K, int epi, void* out, int64_t ldo) {
    vtGemmNT p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = M; p.N = N; p.K = K; p.epi = epi; p.out = out; p.ldo = ldo;
    return p;
}

static int pack_all(vtGatedStack* t, const vtGatedLayerTensors* P, void* ws, vtStream s) {
    const int D = t->c.D, I2 = 2 * t->c.inner, inner = t->c.inner, ip = t->ipad;
    std::vector<vtPackJob> jobs;
    for (int i = 0; i < t->c.depth; ++i) {
        const GLayerBufs& l = t->L[i];
        auto job = [&](const float* w, int N, int K, size_t wb, int64_t ldd, size_t wt, int64_t lddT) {
            vtPackJob q;
            memset(&q, 0, sizeof(q));
            q.w = w; q.N = N; q.K = K; q.wb = WS(void, wb); q.ldd = ldd; q.wt = WS(void, wt); q.lddT = lddT;
            jobs.push_back(q);
        };
        job(P[i].to_qkv_w, 4 * D, D, l.qkv_wb, D, l.qkv_wt, 4 * D);
        job(P[i].out_proj_w, D, D, l.out_wb, D, l.out_wt, D);
        job(P[i].fc1_w, I2, D, l.fc1_wb, D, l.fc1_wt, I2);
        job(P[i].fc2_w, D, inner, l.fc2_wb, ip, l.fc2_wt, D);     // contraction dim padded to ip: pad columns / rows stay zero
    }
    for (size_t j = 0; j < jobs.size(); j += VT_PACK_MAX_GROUP) {
        const int n = (int)(jobs.size() - j < VT_PACK_MAX_GROUP ? jobs.size() - j : VT_PACK_MAX_GROUP);
        TRY(vt_pack_weights_grouped(jobs.data() + j, n, s));
    }
    return VT_OK;
}

extern "C" int vt_gated_stack_forward(vtGatedStack* t, const vtGatedLayerTensors* P, const float* cos_tab, const float* sin_tab, const float* x_in,
                                      void* ws, float* x_out, int32_t repack, vtStream s) {
    VT_CHECK_ARG(t && P && cos_tab && sin_tab && x_in && ws && x_out, "vt_gated_stack_forward: null pointer");
    const vtGatedStackConfig& c = t->c;
    const int M = t->M, D = c.D, I2 = 2 * c.inner, ip = t->ipad;
    const vtRowMap id = {0, 0, 0};
    hipStream_t hs = (hipStream_t)s;
    if (repack) TRY(pack_all(t, P, ws, s));
    if (hipMemcpyAsync(WS(void, t->L[0].x_in), x_in, (size_t)M * D * 4, hipMemcpyDeviceToDevice, hs) != hipSuccess) {
        vt_set_error("vt_gated_stack_forward: copy failed");
        return VT_ERR_LAUNCH;
    }
    for (int i = 0; i < c.depth; ++i) {
        const GLayerBufs& l = t->L[i];
        const float* x = WS(float, l.x_in);
        float* xo = i + 1 < c.depth ? WS(float, t->L[i + 1].x_in) : WS(float, t->x_out);
        TRY(vt_cast_rows(x, id, M, D, WS(void, l.xb), D, s));
        vtGemmNT g = gnt(WS(void, l.xb), D, WS(void, l.qkv_wb), D, M, 4 * D, D, VT_EPI_BF16, WS(void, l.qkvg), 4 * D);
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_qknorm_rope_fwd(WS(void, l.qkvg), M, c.L, c.H, P[i].q_norm_w, P[i].q_norm_b, P[i].k_norm_w, P[i].k_norm_b, 1e-5f, cos_tab, sin_tab,
                               WS(void, l.qkv), s));
        TRY(vt_attention_fwd(WS(void, l.qkv), c.B, c.L, c.H, 64, WS(void, l.o), WS(float, l.lse), s));
        TRY(vt_sigmoid_gate_fwd(WS(void, l.o), WS(void, l.qkvg), M, D, WS(void, l.og), s));
        g = gnt(WS(void, l.og), D, WS(void, l.out_wb), D, M, D, D, VT_EPI_F32, WS(void, l.x1), D);
        g.residual = x; g.ldr = D; g.round_bf16 = 1;
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_layernorm_fwd(WS(float, l.x1), id, P[i].ln_w, P[i].ln_b, 1e-5f, M, D, WS(void, l.y), WS(float, l.mean), WS(float, l.rstd), s));
        g = gnt(WS(void, l.y), D, WS(void, l.fc1_wb), D, M, I2, D, VT_EPI_BF16, WS(void, l.h), I2);
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_geglu_fwd(WS(void, l.h), M, c.inner, WS(void, l.a), ip, s));
        g = gnt(WS(void, l.a), ip, WS(void, l.fc2_wb), ip, M, D, ip, VT_EPI_F32, xo, D);
        g.residual = WS(float, l.x1); g.ldr = D; g.round_bf16 = 1;
        if (i > 0) g.out_scale = (float)(1.0 / sqrt((double)(i + 1)));      // x * (1 / sqrt(i + 1)), transformer.py:88-90
        TRY(vt_gemm_nt(&g, s));
    }
    if (hipMemcpyAsync(x_out, WS(void, t->x_out), (size_t)M * D * 4, hipMemcpyDeviceToDevice, hs) != hipSuccess) {
        vt_set_error("vt_gated_stack_forward: copy failed");
        return VT_ERR_LAUNCH;
    }
    VT_CHECK_LAUNCH("vt_gated_stack_forward");
    return VT_OK;
}

extern "C" int vt_gated_stack_backward(vtGatedStack* t, const vtGatedLayerTensors* P, const float* cos_tab, const float* sin_tab, const float* dy,
                                       void* ws, const vtGatedLayerTensors* G, float* dx, vtStream s) {
    VT_CHECK_ARG(t && P && cos_tab && sin_tab && dy && ws && G && dx, "vt_gated_stack_backward: null pointer");
    const vtGatedStackConfig& c = t->c;
    const int M = t->M, D = c.D, I2 = 2 * c.inner, ip = t->ipad, inner = c.inner;
    const vtRowMap id = {0, 0, 0};
    hipStream_t hs = (hipStream_t)s;
    const float* dcur = dy;                      // gradient w.r.t. the current layer's (rescaled) output
    for (int i = c.depth - 1; i >= 0; --i) {
        const GLayerBufs& l = t->L[i];
        const float scale = i > 0 ? (float)(1.0 / sqrt((double)(i + 1))) : 1.0f;
        // d2 = scale * dout (fp32, residual path) and its bf16 copy (GEMM operand)
        TRY(vt_scale_rows(dcur, scale, M, D, WS(float, t->d2), WS(void, t->gb), s));
        // ffd backward
        vtGemmNT g = gnt(WS(void, t->gb), D, WS(void, l.fc2_wt), D, M, ip, D, VT_EPI_BF16, WS(void, t->da), ip);
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_geglu_bwd(WS(void, t->da), ip, WS(void, l.h), M, inner, WS(void, t->dh), s));
        g = gnt(WS(void, t->dh), I2, WS(void, l.fc1_wt), I2, M, D, I2, VT_EPI_BF16, WS(void, t->dy), D);
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_layernorm_bwd(WS(void, t->dy), WS(float, l.x1), id, P[i].ln_w, WS(float, l.mean), WS(float, l.rstd), WS(float, t->d2), M, D, WS(float, t->dx1),
                             WS(void, t->dx1b), G[i].ln_w, G[i].ln_b, nullptr, WS(void, t->ln_ws), s));
        // attention backward
        g = gnt(WS(void, t->dx1b), D, WS(void, l.out_wt), D, M, D, D, VT_EPI_BF16, WS(void, t->dog), D);
        TRY(vt_gemm_nt(&g, s));
        TRY(vt_sigmoid_gate_bwd(WS(void, t->dog), WS(void, l.o), WS(void, l.qkvg), M, D, WS(void, t->d_o), WS(void, t->dqkvg), s));
        TRY(vt_attention_bwd(WS(void, l.qkv), WS(void, l.o), WS(void, t->d_o), WS(float, l.lse), c.B, c.L, c.H, 64, WS(void, t->dqkv), WS(float, t->delta), s));
        TRY(vt_qknorm_rope_bwd(WS(void, l.qkvg), WS(void, t->dqkv), M, c.L, c.H, P[i].q_norm_w, P[i].k_norm_w, 1e-5f, cos_tab, sin_tab, WS(void, t->dqkvg),
                               G[i].q_norm_w, G[i].q_norm_b, G[i].k_norm_w, G[i].k_norm_b, WS(void, t->qk_ws), s));
        float* dxo = i > 0 ? WS(float, t->L[i].x_in) : dx;   // the layer's saved input is dead once its bf16 copy xb has been used below... (see order)
        // weight gradients of the layer, one grouped launch: dW = dY^T X
        vtGemmTN w[4];
        memset(w, 0, sizeof(w));
        auto tn = [&](vtGemmTN& q, const void* A, int64_t lda, const void* Bm, int64_t ldb, int Pd, int Qd, float* out, int64_t ldo) {
            q.A = A; q.lda = lda; q.B = Bm; q.ldb = ldb; q.M = M; q.P = Pd; q.Q = Qd; q.out = out; q.ldo = ldo; q.p_lim = Pd; q.q_lim = Qd;
        };
        tn(w[0], WS(void, t->gb), D, WS(void, l.a), ip, D, ip, G[i].fc2_w, inner);          // [D, inner] (pad columns not stored: q_lim)
        w[0].q_lim = inner;
        tn(w[1], WS(void, t->dh), I2, WS(void, l.y), D, I2, D, G[i].fc1_w, D);
        tn(w[2], WS(void, t->dx1b), D, WS(void, l.og), D, D, D, G[i].out_proj_w, D);
        tn(w[3], WS(void, t->dqkvg), 4 * D, WS(void, l.xb), D, 4 * D, D, G[i].to_qkv_w, D);
        TRY(vt_gemm_tn_grouped(w, 4, s));
        // input gradient: dx = round_bf16(dqkvg . Wqkv) + dx1
        g = gnt(WS(void, t->dqkvg), 4 * D, WS(void, l.qkv_wt), 4 * D, M, D, 4 * D, VT_EPI_F32, dxo, D);
        g.residual = WS(float, t->dx1); g.ldr = D; g.round_bf16 = 1;
        TRY(vt_gemm_nt(&g, s));
        dcur = dxo;
    }
    (void)hs;
    VT_CHECK_LAUNCH("vt_gated_stack_backward");
    return VT_OK;
}
