// Kernels of the AR consumer (`LARP_AR`, /root/reference/models/larp_ar.py:233-438) that the tokenizer path did
// not already have.  Its matrix products are vt_gemm_nt / vt_gemm_tn_grouped, its training attention vt_attention_causal_*.
//   rmsnorm       RMSNorm (models/norm.py:6-17): y = x * rsqrt(mean(x^2) + eps) * w, fp32 statistics, bf16 output for the next GEMM
//   swiglu        FeedForward (larp_ar.py:122-136): silu(w1 x) * (w3 x) on the packed projection h = [w3 x | w1 x]
//   decode_attn   one new token against the KV cache (larp_ar.py:138-190 with `mask = causal_mask[:, None, input_pos]`): a GEMV-sized
//                 softmax(q K^T / 8) V per (batch, head), memory-bound, one wave per (b, h)
// All HBM-bound single passes, 8- or 16-byte accesses, rounding points of autocast(bf16).
#include "vt_common.h"

namespace {

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
    const float s = 1.0f / (1.0f + __expf(-x));
    return s * (1.0f + x * (1.0f - s));
}

// ---------------------------------------------------------------------------------------------------- RMSNorm
// one wave per row; lane l owns the float2 pieces (j * 64 + l), j < J = dim / 128: every load instruction of the wave is a
// contiguous 512-byte run.  J in {3, 6, 8, 10, 12, 20} <=> dim in {384, 768, 1024, 1280, 1536, 2560} (every llama-abs size).
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int J>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps, int64_t rows,
                                                           bf16_t* __restrict__ y, float* __restrict__ rstd_out) {
    constexpr int dim = J * 128;
    const int lane = threadIdx.x & 63;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const f32x2* xr = (const f32x2*)(x + r * dim);
        f32x2 v[J];
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            v[j] = xr[j * 64 + lane];
            ss = fmaf(v[j][0], v[j][0], ss);
            ss = fmaf(v[j][1], v[j][1], ss);
        }
        ss = wave_sum(ss);
        const float rstd = __builtin_amdgcn_rsqf(ss * (1.0f / dim) + eps);
        if (lane == 0 && rstd_out) rstd_out[r] = rstd;
        bf16x2* yr = (bf16x2*)(y + r * dim);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const f32x2 ww = ((const f32x2*)w)[j * 64 + lane];
            yr[j * 64 + lane] = (bf16x2){f2bf(v[j][0] * rstd * ww[0]), f2bf(v[j][1] * rstd * ww[1])};
        }
    }
}

// dx = rstd * g - x * rstd^3 / dim * sum(x * g) (+ dres), g = w * dy; per-block partial sums of dw = sum_rows dy * x * rstd
template <int J>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ rstd_in, const float* __restrict__ dres, int64_t rows,
                                                           float* __restrict__ dx, bf16_t* __restrict__ dxb, float* __restrict__ dw_part) {
    constexpr int dim = J * 128;
    __shared__ float red[4][dim];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f32x2 dwacc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) dwacc[j] = (f32x2){0.f, 0.f};
    for (int64_t r = (int64_t)blockIdx.x * 4 + wv; r < rows; r += (int64_t)gridDim.x * 4) {
        const f32x2* xr = (const f32x2*)(x + r * dim);
        const bf16x2* dyr = (const bf16x2*)(dy + r * dim);
        const float rstd = rstd_in[r];
        f32x2 xv[J], g[J];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            xv[j] = xr[j * 64 + lane];
            const bf16x2 d = dyr[j * 64 + lane];
            const f32x2 ww = ((const f32x2*)w)[j * 64 + lane];
            const f32x2 dyf = {bf2f(d[0]), bf2f(d[1])};
            g[j] = dyf * ww;
            dot = fmaf(xv[j][0], g[j][0], dot);
            dot = fmaf(xv[j][1], g[j][1], dot);
            dwacc[j] += dyf * xv[j] * rstd;
        }
        dot = wave_sum(dot);
        const float k = dot * rstd * rstd * rstd * (1.0f / dim);
#pragma unroll
        for (int j = 0; j < J; ++j) {
            f32x2 o = g[j] * rstd - xv[j] * k;
            if (dres) o += ((const f32x2*)(dres + r * dim))[j * 64 + lane];
            if (dx) ((f32x2*)(dx + r * dim))[j * 64 + lane] = o;
            if (dxb) ((bf16x2*)(dxb + r * dim))[j * 64 + lane] = (bf16x2){f2bf(o[0]), f2bf(o[1])};
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) ((f32x2*)red[wv])[j * 64 + lane] = dwacc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < dim; c += 256) dw_part[(int64_t)blockIdx.x * dim + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

constexpr int RMS_BLOCKS = 256;   // partial-sum rows of the weight gradient (fixed: the reduction order does not depend on the row count)

// ---------------------------------------------------------------------------------------------------- SwiGLU
struct V8 {
    float v[8];
};
__device__ __forceinline__ V8 ld8(const bf16_t* p) {
    const bf16x8 r = *(const bf16x8*)p;
    V8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = bf2f(r[i]);
    return o;
}
__device__ __forceinline__ void st8(bf16_t* p, const V8& a) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f2bf(a.v[i]);
    *(bf16x8*)p = r;
}

// h [M, 2I] = [w3 x | w1 x]; a = silu(h[:, I:]) * h[:, :I]
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ h, int64_t M, int I, bf16_t* __restrict__ a) {
    const int64_t per_row = I / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const V8 x = ld8(h + row * 2 * I + col), g = ld8(h + row * 2 * I + I + col);
        V8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = round_bf16(silu_f(g.v[i])) * x.v[i];
        st8(a + row * I + col, r);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ da, const bf16_t* __restrict__ h, int64_t M, int I, bf16_t* __restrict__ dh) {
    const int64_t per_row = I / 8, total = M * per_row;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < total; u += (int64_t)gridDim.x * 256) {
        const int64_t row = u / per_row;
        const int col = (int)(u % per_row) * 8;
        const V8 dy = ld8(da + row * I + col), x = ld8(h + row * 2 * I + col), g = ld8(h + row * 2 * I + I + col);
        V8 dx, dg;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            dx.v[i] = dy.v[i] * round_bf16(silu_f(g.v[i]));
            dg.v[i] = round_bf16(dy.v[i] * x.v[i]) * silu_grad_f(g.v[i]);
        }
        st8(dh + row * 2 * I + col, dx);
        st8(dh + row * 2 * I + I + col, dg);
    }
}

// ---------------------------------------------------------------------------------------------------- decode attention (KV cache)
// One new token per sequence against the KV cache (larp_ar.py:138-190 with `mask = causal_mask[:, None, input_pos]`): q [64] of the
// new token, caches k, v [Bmax, H, Lmax, 64] bf16, keys 0..n_keys-1 visible.  HBM-bound (the cache is read once: 256 B per key
// and head), so: one 4-wave workgroup per (b, h), the 8-key groups dealt round-robin to the waves; a wave reads a group as
// 8 rows x 128 B with 16 B per lane (lane = 8 * key_in_group + chunk), i.e. one fully coalesced 1 KB access per instruction.
//   phase 1: scores s = q.k / 8 (8-lane shuffle reduction per key) into LDS, running maximum
//   phase 2: p = exp(s - max) in LDS, sum
//   phase 3: o = sum_k bf16(p_k / sum) v_k : each lane accumulates ITS key slot's contribution to its 8 dims; slots and waves are
//            reduced at the end (fixed order => bit-reproducible)
// STEP variant (the generation loop): the position comes from DEVICE memory (`pos_dev`: number of earlier keys), the new token's
// k and v are taken from the packed projection row qkv[b] = [q | k | v], stored into the caches at `pos` and used from registers --
// no host synchronisation and no separate cache-update launches, so a whole decode step can sit in one hipGraph.
template <bool STEP>
__global__ __launch_bounds__(256) void decode_attn_kernel(const bf16_t* __restrict__ q_or_qkv, bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, int H, int64_t Lmax,
                                                           int n_keys_arg, const int* __restrict__ pos_dev, bf16_t* __restrict__ o) {
    extern __shared__ float probs[];            // [keys rounded up to 32] scores -> probabilities
    __shared__ float red[4][64];
    __shared__ float red_m[4], red_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = lane >> 3, chunk = lane & 7;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int D = H * 64;
    int n_keys = n_keys_arg, pos = -1;
    const bf16_t *qp, *knew = nullptr, *vnew = nullptr;
    if constexpr (STEP) {
        pos = *pos_dev;
        pos = pos < 0 ? 0 : pos;
        const bf16_t* row = q_or_qkv + (int64_t)b * 3 * D + h * 64;
        qp = row, knew = row + D, vnew = row + 2 * D;
        if (pos >= Lmax) pos = (int)Lmax - 1;       // caller error (checked on the host where the position is known); stay in bounds
        n_keys = pos + 1;
    } else {
        qp = q_or_qkv + ((int64_t)b * H + h) * 64;
    }
    bf16_t* kp = kc + ((int64_t)b * H + h) * Lmax * 64;
    bf16_t* vp = vc + ((int64_t)b * H + h) * Lmax * 64;
    if (STEP && wave == 0 && lane < 16) {           // the new token enters the cache
        const int c = lane & 7;
        if (lane < 8) *(bf16x8*)(kp + (int64_t)pos * 64 + 8 * c) = *(const bf16x8*)(knew + 8 * c);
        else *(bf16x8*)(vp + (int64_t)pos * 64 + 8 * c) = *(const bf16x8*)(vnew + 8 * c);
    }
    const V8 qv = ld8(qp + 8 * chunk);
    const int groups = (n_keys + 7) / 8;
    // phase 1
    float mx = -__builtin_inff();
    for (int g0 = wave; g0 < groups; g0 += 16) {         // 4 groups of 8 keys per wave in flight
        V8 kv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (g0 + 4 * u) * 8 + slot;
            const bf16_t* src = (STEP && key == pos) ? knew : kp + (int64_t)(key < n_keys ? key : 0) * 64;
            kv[u] = ld8(src + 8 * chunk);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (g0 + 4 * u) * 8 + slot;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(qv.v[i], kv[u].v[i], s);
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s = key < n_keys ? s * 0.125f : -__builtin_inff();
            if (chunk == 0 && g0 + 4 * u < groups) probs[key] = s;
            mx = fmaxf(mx, s);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) red_m[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    // phase 2
    float sum = 0.f;
    for (int k = tid; k < groups * 8; k += 256) {
        const float p = __expf(probs[k] - mx);     // exp(-inf) = 0 for the padded keys
        probs[k] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    if (lane == 0) red_s[wave] = sum;
    __syncthreads();
    sum = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    const float inv = 1.0f / sum;
    // phase 3
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int g0 = wave; g0 < groups; g0 += 16) {
        V8 vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (g0 + 4 * u) * 8 + slot;
            const bf16_t* src = (STEP && key == pos) ? vnew : vp + (int64_t)(key < n_keys ? key : 0) * 64;
            vv[u] = ld8(src + 8 * chunk);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = (g0 + 4 * u) * 8 + slot;
            const float pr = key < n_keys ? round_bf16(probs[key] * inv) : 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(pr, vv[u].v[i], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        acc[i] += __shfl_xor(acc[i], 8);
        acc[i] += __shfl_xor(acc[i], 16);
        acc[i] += __shfl_xor(acc[i], 32);
    }
    if (slot == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) red[wave][8 * chunk + i] = acc[i];
    }
    __syncthreads();
    if (tid < 64) {
        const float r = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        o[((int64_t)b * H + h) * 64 + tid] = f2bf(r);
    }
}


// ---------------------------------------------------------------------------------------------------- decode step: RMSNorm + Linear [+ SwiGLU]
// One node of the generation graph instead of two or three: y = Linear(RMSNorm(x)) for the M <= 64 token rows of a decode step.
// A workgroup owns 16 weight rows (as gemm_nt_skinny_kernel); it first computes the M row scales itself (the rows are a few KB,
// L2-resident; same summation order as rmsnorm_fwd_kernel, so the result is bit-identical to the two-kernel path), then forms the
// bf16 operand fragments bf16(x * rstd * w) on the fly.  MODE 0: bf16 output (wqkv).  MODE 1: the weight is [w3 ; w1] interleaved in
// slabs of 8 + 8 rows, the epilogue applies SwiGLU with autocast's rounding points and writes a [M, N / 2].  MODE 2: fp32 output of
// the bf16-rounded value (the LM head's logits).
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int MT, int MODE, bool NORM>
__global__ __launch_bounds__(256) void decode_norm_linear_kernel(const void* __restrict__ xin, const float* __restrict__ nw, float eps, const bf16_t* __restrict__ W,
                                                                  int M, int N, int K, void* __restrict__ out, int64_t ldo) {
    __shared__ float red[4][MT * 16 * 16];
    __shared__ float rs[MT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int fr = lane & 15, fq = lane >> 4;
    if constexpr (NORM) {       // the row scales: this wave's MT * 4 rows side by side (independent loads), each in rmsnorm_fwd_kernel's order
        const float* x = (const float*)xin;
        constexpr int RW = MT * 4;
        float ss[RW];
        const f32x2* xr[RW];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int m = wave + 4 * r;
            xr[r] = (const f32x2*)(x + (int64_t)(m < M ? m : M - 1) * K);
            ss[r] = 0.f;
        }
        for (int j = 0; j < K / 128; ++j) {
            f32x2 v[RW];
#pragma unroll
            for (int r = 0; r < RW; ++r) v[r] = xr[r][j * 64 + lane];
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                ss[r] = fmaf(v[r][0], v[r][0], ss[r]);
                ss[r] = fmaf(v[r][1], v[r][1], ss[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const float t = wave_sum(ss[r]);
            if (lane == 0) rs[wave + 4 * r] = __builtin_amdgcn_rsqf(t * (1.0f / K) + eps);
        }
        __syncthreads();
    }
    int wrow = n0 + fr;
    wrow = wrow < N ? wrow : N - 1;
    const bf16_t* wp = W + (int64_t)wrow * K + 8 * fq;
    int64_t xoff[MT];
    float rstd[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int r = t * 16 + fr;
        rstd[t] = NORM ? rs[r] : 1.f;
        r = r < M ? r : M - 1;
        xoff[t] = (int64_t)r * K + 8 * fq;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = K / 32;
    constexpr int U = NORM ? (MT <= 2 ? 4 : 2) : 4;      // k-steps per wave in flight
    for (int s0 = wave; s0 < steps; s0 += 4 * U) {
        bf16x8 wf[U], xf[U][MT];
        if constexpr (NORM) {
            const float* x = (const float*)xin;
            f32x4v xa[U][MT], xb[U][MT], na[U], nb[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = s0 + 4 * u;
                const int kk = (ks < steps ? ks : s0) * 32;
                wf[u] = *(const bf16x8*)(wp + kk);
                na[u] = *(const f32x4v*)(nw + kk + 8 * fq);
                nb[u] = *(const f32x4v*)(nw + kk + 8 * fq + 4);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    xa[u][t] = *(const f32x4v*)(x + xoff[t] + kk);
                    xb[u][t] = *(const f32x4v*)(x + xoff[t] + kk + 4);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        xf[u][t][i] = f2bf(xa[u][t][i] * rstd[t] * na[u][i]);
                        xf[u][t][4 + i] = f2bf(xb[u][t][i] * rstd[t] * nb[u][i]);
                    }
        } else {
            const bf16_t* x = (const bf16_t*)xin;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ks = s0 + 4 * u;
                const int kk = (ks < steps ? ks : s0) * 32;
                wf[u] = *(const bf16x8*)(wp + kk);
#pragma unroll
                for (int t = 0; t < MT; ++t) xf[u][t] = *(const bf16x8*)(x + xoff[t] + kk);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (s0 + 4 * u < steps) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[u], xf[u][t], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *(f32x4*)(&red[wave][(t * 16 + fr) * 16 + 4 * fq]) = acc[t];
    __syncthreads();
    const int m = tid >> 2, g = tid & 3;
    if (m >= M || m >= MT * 16) return;
    auto total = [&](int c4) {
        return (*(const f32x4*)(&red[0][m * 16 + 4 * c4]) + *(const f32x4*)(&red[1][m * 16 + 4 * c4])) +
               (*(const f32x4*)(&red[2][m * 16 + 4 * c4]) + *(const f32x4*)(&red[3][m * 16 + 4 * c4]));
    };
    if constexpr (MODE == 1) {
        if (g < 2) {            // hidden units 4g..4g+3 of this slab: value rows 4g.., gate rows 8 + 4g..
            const f32x4 xv = total(g), gv = total(2 + g);
            bf16x4 r;
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = f2bf(round_bf16(silu_f(round_bf16(gv[i]))) * round_bf16(xv[i]));
            *(bf16x4*)((bf16_t*)out + (int64_t)m * ldo + blockIdx.x * 8 + 4 * g) = r;
        }
    } else {
        const f32x4 v = total(g);
        const int n = n0 + 4 * g;
        if (n < N) {
            if constexpr (MODE == 0) *(bf16x4*)((bf16_t*)out + (int64_t)m * ldo + n) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            else *(f32x4*)((float*)out + (int64_t)m * ldo + n) = (f32x4){round_bf16(v[0]), round_bf16(v[1]), round_bf16(v[2]), round_bf16(v[3])};
        }
    }
}

int grid_for(int64_t units) {
    const int64_t b = (units + 255) / 256;
    return (int)(b < 1 ? 1 : b > 4096 ? 4096 : b);
}
}  // namespace

#define RMS_DISPATCH(KERNEL, ...)                                                              \
    switch (dim / 128) {                                                                       \
        case 3: hipLaunchKernelGGL(KERNEL<3>, __VA_ARGS__); break;                             \
        case 6: hipLaunchKernelGGL(KERNEL<6>, __VA_ARGS__); break;                             \
        case 8: hipLaunchKernelGGL(KERNEL<8>, __VA_ARGS__); break;                             \
        case 10: hipLaunchKernelGGL(KERNEL<10>, __VA_ARGS__); break;                           \
        case 12: hipLaunchKernelGGL(KERNEL<12>, __VA_ARGS__); break;                           \
        default: hipLaunchKernelGGL(KERNEL<20>, __VA_ARGS__); break;                           \
    }

static bool rms_dim_ok(int dim) { return dim == 384 || dim == 768 || dim == 1024 || dim == 1280 || dim == 1536 || dim == 2560; }

extern "C" int vt_rmsnorm_fwd(const float* x, const float* w, float eps, int64_t rows, int32_t dim, void* y_bf16, float* rstd, vtStream stream) {
    VT_CHECK_ARG(x && w && y_bf16 && rows > 0, "vt_rmsnorm_fwd: null pointer");
    VT_CHECK_ARG(rms_dim_ok(dim), "vt_rmsnorm_fwd: width %d unsupported (384, 768, 1024, 1280, 1536, 2560: the llama-abs sizes)", dim);
    const int grid = (int)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
    RMS_DISPATCH(rmsnorm_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, w, eps, rows, (bf16_t*)y_bf16, rstd)
    VT_CHECK_LAUNCH("vt_rmsnorm_fwd");
    return VT_OK;
}

extern "C" size_t vt_rmsnorm_bwd_workspace_bytes(int32_t dim) { return (size_t)RMS_BLOCKS * dim * sizeof(float); }

extern "C" int vt_rmsnorm_bwd(const void* dy_bf16, const float* x, const float* w, const float* rstd, const float* dres, int64_t rows, int32_t dim,
                              float* dx, void* dx_bf16, float* dw, void* workspace, vtStream stream) {
    VT_CHECK_ARG(dy_bf16 && x && w && rstd && (dx || dx_bf16) && dw && workspace && rows > 0, "vt_rmsnorm_bwd: null pointer");
    VT_CHECK_ARG(rms_dim_ok(dim), "vt_rmsnorm_bwd: width %d unsupported", dim);
    float* part = (float*)workspace;
    RMS_DISPATCH(rmsnorm_bwd_kernel, dim3(RMS_BLOCKS), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy_bf16, x, w, rstd, dres, rows, dx, (bf16_t*)dx_bf16, part)
    VT_CHECK_LAUNCH("vt_rmsnorm_bwd");
    return vt_sum_slabs(part, RMS_BLOCKS, (int64_t)dim, dim, dw, stream);
}

extern "C" int vt_swiglu_fwd(const void* h, int64_t M, int32_t I, void* a, vtStream stream) {
    VT_CHECK_ARG(h && a && M > 0 && I > 0 && I % 8 == 0, "vt_swiglu_fwd: null pointer or I %% 8 != 0");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)h, M, I, (bf16_t*)a);
    VT_CHECK_LAUNCH("vt_swiglu_fwd");
    return VT_OK;
}

extern "C" int vt_swiglu_bwd(const void* da, const void* h, int64_t M, int32_t I, void* dh, vtStream stream) {
    VT_CHECK_ARG(da && h && dh && M > 0 && I > 0 && I % 8 == 0, "vt_swiglu_bwd: null pointer or I %% 8 != 0");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(M * (I / 8))), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)da, (const bf16_t*)h, M, I, (bf16_t*)dh);
    VT_CHECK_LAUNCH("vt_swiglu_bwd");
    return VT_OK;
}

extern "C" int vt_decode_attention(const void* q, const void* k_cache, const void* v_cache, int32_t B, int32_t H, int64_t Lmax, int32_t n_keys, void* o,
                                   vtStream stream) {
    VT_CHECK_ARG(q && k_cache && v_cache && o && B > 0 && H > 0 && n_keys > 0 && n_keys <= Lmax, "vt_decode_attention: bad arguments (1 <= n_keys <= Lmax)");
    VT_CHECK_ARG(n_keys <= 12288, "vt_decode_attention: n_keys %d > 12288", n_keys);
    const size_t lds = (size_t)((n_keys + 7) / 8 * 8) * sizeof(float);
    hipLaunchKernelGGL(decode_attn_kernel<false>, dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)q, (bf16_t*)k_cache, (bf16_t*)v_cache, H, Lmax,
                       n_keys, (const int*)nullptr, (bf16_t*)o);
    VT_CHECK_LAUNCH("vt_decode_attention");
    return VT_OK;
}

extern "C" int vt_decode_attention_step(const void* qkv, void* k_cache, void* v_cache, int32_t B, int32_t H, int64_t Lmax, const int32_t* pos_dev, void* o,
                                        vtStream stream) {
    VT_CHECK_ARG(qkv && k_cache && v_cache && pos_dev && o && B > 0 && H > 0 && Lmax > 0, "vt_decode_attention_step: bad arguments");
    VT_CHECK_ARG(Lmax <= 12288, "vt_decode_attention_step: Lmax %ld > 12288", (long)Lmax);
    const size_t lds = (size_t)((Lmax + 7) / 8 * 8) * sizeof(float);      // the position is only known on the device: size for the whole cache
    hipLaunchKernelGGL(decode_attn_kernel<true>, dim3(B * H), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)k_cache, (bf16_t*)v_cache, H, Lmax, 0,
                       (const int*)pos_dev, (bf16_t*)o);
    VT_CHECK_LAUNCH("vt_decode_attention_step");
    return VT_OK;
}

extern "C" int vt_decode_norm_linear(const void* x, const float* norm_w, float eps, const void* W_bf16, int32_t M, int32_t N, int32_t K, int32_t mode, void* out,
                                     int64_t ldo, vtStream stream) {
    VT_CHECK_ARG(x && W_bf16 && out, "vt_decode_norm_linear: null pointer");
    VT_CHECK_ARG(M > 0 && M <= 64 && N > 0 && N % 16 == 0 && K > 0 && K % 128 == 0, "vt_decode_norm_linear: M=%d (1..64) N=%d (%%16) K=%d (%%128)", M, N, K);
    VT_CHECK_ARG(mode >= 0 && mode <= 2 && ldo % 4 == 0 && ldo >= (mode == 1 ? N / 2 : N), "vt_decode_norm_linear: mode %d / ldo %ld", mode, (long)ldo);
    VT_CHECK_ARG((((uintptr_t)x | (uintptr_t)norm_w | (uintptr_t)W_bf16 | (uintptr_t)out) & 15) == 0, "vt_decode_norm_linear: pointers must be 16-byte aligned");
    const dim3 grid(N / 16), block(256);
    hipStream_t s = (hipStream_t)stream;
    const int mt = (M + 15) / 16;
#define VT_DNL(MT_, MODE_, NORM_) hipLaunchKernelGGL((decode_norm_linear_kernel<MT_, MODE_, NORM_>), grid, block, 0, s, x, norm_w, eps, (const bf16_t*)W_bf16, M, N, K, out, ldo)
#define VT_DNL_M(MODE_, NORM_) switch (mt) { case 1: VT_DNL(1, MODE_, NORM_); break; case 2: VT_DNL(2, MODE_, NORM_); break; case 3: VT_DNL(3, MODE_, NORM_); break; default: VT_DNL(4, MODE_, NORM_); break; }
    if (norm_w) {
        if (mode == 0) { VT_DNL_M(0, true) } else if (mode == 1) { VT_DNL_M(1, true) } else { VT_DNL_M(2, true) }
    } else {
        if (mode == 0) { VT_DNL_M(0, false) } else if (mode == 1) { VT_DNL_M(1, false) } else { VT_DNL_M(2, false) }
    }
#undef VT_DNL_M
#undef VT_DNL
    VT_CHECK_LAUNCH("vt_decode_norm_linear");
    return VT_OK;
}
