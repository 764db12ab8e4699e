// Elementwise pieces of the reference's TiTok-style transformer block (models/model_new/base/transformer.py) for gfx950:
//   Attn.forward :45-63   q, k, v, gate = to_qkv(x).chunk(4); q, k = LayerNorm_hd(q), LayerNorm_hd(k); rotary(q), rotary(k);
//                         x = flash_attn(q, k, v); x = x * sigmoid(gate); out_proj(x)
//   ffd / GEGLU  :11-29   LayerNorm -> Linear(D, 2 I) -> gelu(gate) * x -> Linear(I, D)
//   apply_rotary_emb      models/model_new/base/rope.py:18-24 (adjacent pairs as complex numbers times freqs_cis[pos])
// The GEMMs, the LayerNorm over D and the attention itself are the kernels of the LARP path; what is new here is
// HBM-bound glue, each a single pass with 16-byte accesses:
//   qknorm_rope   reads q|k|v of the packed [M, 4D] projection, writes the packed [M, 3D] operand of vt_attention_*
//   sigmoid_gate  o * sigmoid(gate), gate read in place from columns 3D..4D of the projection
//   geglu         gelu(h[:, I:]) * h[:, :I]
// and their backward passes, which write straight into the [M, 4D] / [M, 2I] gradient of the projection so no torch
// cat/chunk copies are needed.  Rounding points follow autocast(bf16): LayerNorm output, rotary output, sigmoid, gelu and
// every product are rounded to bf16 where the reference materialises a bf16 tensor; statistics and products are fp32.
#include "vt_common.h"

namespace {
constexpr int HD = 64;       // head_dim of every model size (models/model_new/base/utils.py:6)
constexpr int VPB = 32;      // head vectors per 256-thread block (8 lanes x 8 elements = one head vector)
constexpr int NBLK = 512;    // blocks per operand in the backward (partial sums: [NBLK, 2, 2, 64] fp32)

__device__ __forceinline__ float sum8(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    return v;
}

struct Vec8 {
    float v[8];
};
__device__ __forceinline__ Vec8 load8(const bf16_t* p) {
    const bf16x8 r = *(const bf16x8*)p;
    Vec8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = bf2f(r[i]);
    return o;
}
__device__ __forceinline__ void store8(bf16_t* p, const Vec8& a) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f2bf(a.v[i]);
    *(bf16x8*)p = r;
}

// LayerNorm statistics of one 64-vector spread over 8 lanes (biased variance, two passes in registers)
__device__ __forceinline__ void ln_stats(const Vec8& x, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x.v[i];
    mean = sum8(s) * (1.0f / HD);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (x.v[i] - mean) * (x.v[i] - mean);
    rstd = 1.0f / sqrtf(sum8(q) * (1.0f / HD) + eps);
}

// grid (blocks, 3): y = 0 -> q, 1 -> k, 2 -> v (copy)
__global__ __launch_bounds__(256) void qknorm_rope_fwd_kernel(const bf16_t* __restrict__ qkvg, int64_t M, int L, int H, const float* __restrict__ q_w,
                                                               const float* __restrict__ q_b, const float* __restrict__ k_w,
                                                               const float* __restrict__ k_b, float eps, const float* __restrict__ cs,
                                                               const float* __restrict__ sn, bf16_t* __restrict__ out) {
    const int which = blockIdx.y, lane = threadIdx.x & 7;
    const int64_t D = (int64_t)H * HD, nvec = M * H;
    const float* w = which == 0 ? q_w : k_w;
    const float* b = which == 0 ? q_b : k_b;
    float wr[8], br[8];
    if (which < 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[i] = w[lane * 8 + i], br[i] = b[lane * 8 + i];
    }
    for (int64_t vec = (int64_t)blockIdx.x * VPB + (threadIdx.x >> 3); vec < nvec; vec += (int64_t)gridDim.x * VPB) {
        const int64_t row = vec / H;
        const int head = (int)(vec % H);
        const bf16_t* src = qkvg + row * 4 * D + which * D + head * HD + lane * 8;
        bf16_t* dst = out + row * 3 * D + which * D + head * HD + lane * 8;
        if (which == 2) {
            *(bf16x8*)dst = *(const bf16x8*)src;
            continue;
        }
        Vec8 x = load8(src);
        float mean, rstd;
        ln_stats(x, eps, mean, rstd);
        const int pos = (int)(row % L);
        const f32x4 c = *(const f32x4*)(cs + (int64_t)pos * (HD / 2) + lane * 4);
        const f32x4 s = *(const f32x4*)(sn + (int64_t)pos * (HD / 2) + lane * 4);
        Vec8 y;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = round_bf16((x.v[2 * j] - mean) * rstd * wr[2 * j] + br[2 * j]);
            const float bb = round_bf16((x.v[2 * j + 1] - mean) * rstd * wr[2 * j + 1] + br[2 * j + 1]);
            y.v[2 * j] = a * c[j] - bb * s[j];
            y.v[2 * j + 1] = a * s[j] + bb * c[j];
        }
        store8(dst, y);
    }
}

// grid (NBLK, 3).  part: [NBLK, 2(which), 2(w|b), 64]
__global__ __launch_bounds__(256) void qknorm_rope_bwd_kernel(const bf16_t* __restrict__ qkvg, const bf16_t* __restrict__ dqkv, int64_t M, int L, int H,
                                                               const float* __restrict__ q_w, const float* __restrict__ k_w, float eps,
                                                               const float* __restrict__ cs, const float* __restrict__ sn,
                                                               bf16_t* __restrict__ dqkvg, float* __restrict__ part) {
    __shared__ float red[2][VPB][HD];
    const int which = blockIdx.y, lane = threadIdx.x & 7, vslot = threadIdx.x >> 3;
    const int64_t D = (int64_t)H * HD, nvec = M * H;
    float wr[8], aw[8], ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) aw[i] = 0.f, ab[i] = 0.f, wr[i] = which < 2 ? (which == 0 ? q_w : k_w)[lane * 8 + i] : 0.f;
    for (int64_t vec = (int64_t)blockIdx.x * VPB + vslot; vec < nvec; vec += (int64_t)gridDim.x * VPB) {
        const int64_t row = vec / H;
        const int head = (int)(vec % H);
        const bf16_t* gsrc = dqkv + row * 3 * D + which * D + head * HD + lane * 8;
        bf16_t* dst = dqkvg + row * 4 * D + which * D + head * HD + lane * 8;
        if (which == 2) {
            *(bf16x8*)dst = *(const bf16x8*)gsrc;
            continue;
        }
        const Vec8 x = load8(qkvg + row * 4 * D + which * D + head * HD + lane * 8);
        const Vec8 gy = load8(gsrc);
        float mean, rstd;
        ln_stats(x, eps, mean, rstd);
        const int pos = (int)(row % L);
        const f32x4 c = *(const f32x4*)(cs + (int64_t)pos * (HD / 2) + lane * 4);
        const f32x4 s = *(const f32x4*)(sn + (int64_t)pos * (HD / 2) + lane * 4);
        // transpose of the rotation, rounded to bf16 (the gradient of a bf16 tensor), then LayerNorm backward in fp32
        float g[8], xh[8], m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g[2 * j] = round_bf16(gy.v[2 * j] * c[j] + gy.v[2 * j + 1] * s[j]);
            g[2 * j + 1] = round_bf16(gy.v[2 * j + 1] * c[j] - gy.v[2 * j] * s[j]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xh[i] = (x.v[i] - mean) * rstd;
            aw[i] += g[i] * xh[i];
            ab[i] += g[i];
            m1 += g[i] * wr[i];
            m2 += g[i] * wr[i] * xh[i];
        }
        m1 = sum8(m1) * (1.0f / HD);
        m2 = sum8(m2) * (1.0f / HD);
        Vec8 dx;
#pragma unroll
        for (int i = 0; i < 8; ++i) dx.v[i] = rstd * (g[i] * wr[i] - m1 - xh[i] * m2);
        store8(dst, dx);
    }
    if (which == 2) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) red[0][vslot][lane * 8 + i] = aw[i], red[1][vslot][lane * 8 + i] = ab[i];
    __syncthreads();
    if (threadIdx.x < 2 * HD) {
        const int wb = threadIdx.x >> 6, e = threadIdx.x & 63;
        float t = 0.f;
#pragma unroll 8
        for (int v = 0; v < VPB; ++v) t += red[wb][v][e];
        part[(((int64_t)blockIdx.x * 2 + which) * 2 + wb) * HD + e] = t;
    }
}

// one wave per output element (which, w|b, e): lane l adds partials l, l + 64, ... in order, then a fixed butterfly
__global__ __launch_bounds__(256) void qknorm_reduce_kernel(const float* __restrict__ part, int nblk, float* dq_w, float* dq_b, float* dk_w, float* dk_b) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;   // t in [0, 256)
    float s = 0.f;
    for (int b = lane; b < nblk; b += 64) s += part[(int64_t)b * 256 + t];
    s = wave_sum(s);
    float* dst = (t >> 6) == 0 ? dq_w : (t >> 6) == 1 ? dq_b : (t >> 6) == 2 ? dk_w : dk_b;
    if (dst && lane == 0) dst[t & 63] = s;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__global__ __launch_bounds__(256) void gate_fwd_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ qkvg, int64_t M, int D, bf16_t* __restrict__ og) {
    const int64_t per_row = D / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const Vec8 a = load8(o + row * D + col), g = load8(qkvg + row * 4 * D + 3 * (int64_t)D + col);
        Vec8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = a.v[i] * round_bf16(sigmoidf_(g.v[i]));
        store8(og + row * D + col, r);
    }
}

__global__ __launch_bounds__(256) void gate_bwd_kernel(const bf16_t* __restrict__ dog, const bf16_t* __restrict__ o, const bf16_t* __restrict__ qkvg, int64_t M,
                                                        int D, bf16_t* __restrict__ d_o, bf16_t* __restrict__ dqkvg) {
    const int64_t per_row = D / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const Vec8 dy = load8(dog + row * D + col), a = load8(o + row * D + col), g = load8(qkvg + row * 4 * D + 3 * (int64_t)D + col);
        Vec8 da, dg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float sg = round_bf16(sigmoidf_(g.v[i]));
            da.v[i] = dy.v[i] * sg;
            dg.v[i] = round_bf16(dy.v[i] * a.v[i]) * ((1.0f - sg) * sg);
        }
        store8(d_o + row * D + col, da);
        store8(dqkvg + row * 4 * D + 3 * (int64_t)D + col, dg);
    }
}

__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* __restrict__ h, int64_t M, int I, bf16_t* __restrict__ a, int64_t lda) {
    const int64_t per_row = I / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const Vec8 x = load8(h + row * 2 * I + col), g = load8(h + row * 2 * I + I + col);
        Vec8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = round_bf16(gelu_erf(g.v[i])) * x.v[i];
        store8(a + row * lda + col, r);
    }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* __restrict__ da, int64_t lda, const bf16_t* __restrict__ h, int64_t M, int I,
                                                         bf16_t* __restrict__ dh) {
    const int64_t per_row = I / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const Vec8 dy = load8(da + row * lda + col), x = load8(h + row * 2 * I + col), g = load8(h + row * 2 * I + I + col);
        Vec8 dx, dg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dx.v[i] = dy.v[i] * round_bf16(gelu_erf(g.v[i]));
            dg.v[i] = round_bf16(dy.v[i] * x.v[i]) * gelu_erf_grad(g.v[i]);
        }
        store8(dh + row * 2 * I + col, dx);
        store8(dh + row * 2 * I + I + col, dg);
    }
}

int grid_for(int64_t units) {
    const int64_t b = (units + 255) / 256;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}
bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }
}  // namespace

extern "C" int vt_qknorm_rope_fwd(const void* qkvg, int64_t M, int32_t L, int32_t H, const float* q_w, const float* q_b, const float* k_w,
                                  const float* k_b, float eps, const float* cos_tab, const float* sin_tab, void* qkv_out, vtStream stream) {
    VT_CHECK_ARG(qkvg && qkv_out && q_w && q_b && k_w && k_b && cos_tab && sin_tab, "vt_qknorm_rope_fwd: null pointer");
    VT_CHECK_ARG(M > 0 && L > 0 && H > 0 && M % L == 0, "vt_qknorm_rope_fwd: need M = B * L rows, H heads of 64");
    VT_CHECK_ARG(aligned16(qkvg) && aligned16(qkv_out) && aligned16(cos_tab) && aligned16(sin_tab), "vt_qknorm_rope_fwd: buffers must be 16-byte aligned");
    const int64_t nvec = M * H;
    const int gx = (int)((nvec + VPB - 1) / VPB < 2048 ? (nvec + VPB - 1) / VPB : 2048);
    hipLaunchKernelGGL(qknorm_rope_fwd_kernel, dim3(gx, 3), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkvg, M, L, H, q_w, q_b, k_w, k_b, eps,
                       cos_tab, sin_tab, (bf16_t*)qkv_out);
    VT_CHECK_LAUNCH("vt_qknorm_rope_fwd");
    return VT_OK;
}

extern "C" size_t vt_qknorm_rope_bwd_workspace_bytes(void) { return (size_t)NBLK * 256 * sizeof(float); }

extern "C" int vt_qknorm_rope_bwd(const void* qkvg, const void* dqkv, int64_t M, int32_t L, int32_t H, const float* q_w, const float* k_w, float eps,
                                  const float* cos_tab, const float* sin_tab, void* dqkvg, float* dq_w, float* dq_b, float* dk_w, float* dk_b,
                                  void* workspace, vtStream stream) {
    VT_CHECK_ARG(qkvg && dqkv && dqkvg && q_w && k_w && cos_tab && sin_tab && workspace, "vt_qknorm_rope_bwd: null pointer");
    VT_CHECK_ARG(M > 0 && L > 0 && H > 0 && M % L == 0, "vt_qknorm_rope_bwd: need M = B * L rows, H heads of 64");
    VT_CHECK_ARG(aligned16(qkvg) && aligned16(dqkv) && aligned16(dqkvg) && aligned16(cos_tab) && aligned16(sin_tab) && aligned16(workspace),
                 "vt_qknorm_rope_bwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(qknorm_rope_bwd_kernel, dim3(NBLK, 3), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkvg, (const bf16_t*)dqkv, M, L, H, q_w, k_w,
                       eps, cos_tab, sin_tab, (bf16_t*)dqkvg, (float*)workspace);
    VT_CHECK_LAUNCH("vt_qknorm_rope_bwd");
    hipLaunchKernelGGL(qknorm_reduce_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, NBLK, dq_w, dq_b, dk_w, dk_b);
    VT_CHECK_LAUNCH("vt_qknorm_rope_bwd(reduce)");
    return VT_OK;
}

extern "C" int vt_sigmoid_gate_fwd(const void* o, const void* qkvg, int64_t M, int32_t D, void* og, vtStream stream) {
    VT_CHECK_ARG(o && qkvg && og && M > 0 && D > 0 && D % 8 == 0, "vt_sigmoid_gate_fwd: null pointer or D %% 8 != 0");
    VT_CHECK_ARG(aligned16(o) && aligned16(qkvg) && aligned16(og), "vt_sigmoid_gate_fwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(M * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)o, (const bf16_t*)qkvg, M, D, (bf16_t*)og);
    VT_CHECK_LAUNCH("vt_sigmoid_gate_fwd");
    return VT_OK;
}

extern "C" int vt_sigmoid_gate_bwd(const void* dog, const void* o, const void* qkvg, int64_t M, int32_t D, void* d_o, void* dqkvg, vtStream stream) {
    VT_CHECK_ARG(dog && o && qkvg && d_o && dqkvg && M > 0 && D > 0 && D % 8 == 0, "vt_sigmoid_gate_bwd: null pointer or D %% 8 != 0");
    VT_CHECK_ARG(aligned16(dog) && aligned16(o) && aligned16(qkvg) && aligned16(d_o) && aligned16(dqkvg), "vt_sigmoid_gate_bwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(M * (D / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dog, (const bf16_t*)o,
                       (const bf16_t*)qkvg, M, D, (bf16_t*)d_o, (bf16_t*)dqkvg);
    VT_CHECK_LAUNCH("vt_sigmoid_gate_bwd");
    return VT_OK;
}

extern "C" int vt_geglu_fwd(const void* h, int64_t M, int32_t I, void* a, int64_t lda, vtStream stream) {
    VT_CHECK_ARG(h && a && M > 0 && I > 0 && I % 8 == 0 && lda >= I && lda % 8 == 0, "vt_geglu_fwd: null pointer, I %% 8 != 0 or bad lda");
    VT_CHECK_ARG(aligned16(h) && aligned16(a), "vt_geglu_fwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, M, I, (bf16_t*)a, lda);
    VT_CHECK_LAUNCH("vt_geglu_fwd");
    return VT_OK;
}

extern "C" int vt_geglu_bwd(const void* da, int64_t lda, const void* h, int64_t M, int32_t I, void* dh, vtStream stream) {
    VT_CHECK_ARG(da && h && dh && M > 0 && I > 0 && I % 8 == 0 && lda >= I && lda % 8 == 0, "vt_geglu_bwd: null pointer, I %% 8 != 0 or bad lda");
    VT_CHECK_ARG(aligned16(da) && aligned16(h) && aligned16(dh), "vt_geglu_bwd: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)da, lda, (const bf16_t*)h, M, I, (bf16_t*)dh);
    VT_CHECK_LAUNCH("vt_geglu_bwd");
    return VT_OK;
}


// ------------------------------------------------------------------------------------------------ scaled copy of gradient rows
namespace {
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ src, float scale, int64_t n4, float* __restrict__ dst, bf16_t* __restrict__ dstb) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = ((const f32x4*)src)[i];
        v *= scale;
        if (dst) ((f32x4*)dst)[i] = v;
        if (dstb) ((bf16x4*)dstb)[i] = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
    }
}
}  // namespace

extern "C" int vt_scale_rows(const float* src, float scale, int64_t rows, int32_t dim, float* dst_f32, void* dst_bf16, vtStream stream) {
    VT_CHECK_ARG(src && (dst_f32 || dst_bf16) && rows > 0 && dim > 0 && dim % 4 == 0, "vt_scale_rows: bad arguments (dim %% 4 == 0)");
    VT_CHECK_ARG((((uintptr_t)src | (uintptr_t)dst_f32) & 15) == 0 && ((uintptr_t)dst_bf16 & 7) == 0, "vt_scale_rows: unaligned pointer");
    const int64_t n4 = rows * dim / 4;
    const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, scale, n4, dst_f32, (bf16_t*)dst_bf16);
    VT_CHECK_LAUNCH("vt_scale_rows");
    return VT_OK;
}
