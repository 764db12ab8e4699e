// Fused epilogues of the NT GEMM, shared by the 128x128 and 192x192 tile kernels.  One call handles the four
// consecutive output columns n..n+3 of output row m that one lane owns after the swapped (B x A) MFMA.
// Interior tiles (n + 3 < N) take the vector path: 16-B loads of bias / residual / rowmod / aux, one 8- or
// 16-B store per output.  Only the last, ragged column group of a matrix takes the scalar path.
#pragma once
#include "vt_common.h"

template <int EPI>
__device__ __forceinline__ void nt_epilogue(const vtGemmNT& p, const RowMap& omap, int m, int n, const f32x4& acc4) {
    const int64_t orow = (EPI == VT_EPI_F32) ? omap(m) : (int64_t)m;
    if (n + 3 < p.N) {
        f32x4 v = acc4;
        if (p.bias) v += *(const f32x4*)(p.bias + n);
        if constexpr (EPI == VT_EPI_BF16) {
            *(bf16x4*)((bf16_t*)p.out + orow * p.ldo + n) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        } else if constexpr (EPI == VT_EPI_BF16_GELU) {
            // GELU of the bf16-rounded pre-activation, evaluated in fp32 (autocast order)
            const bf16x4 u = {f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            *(bf16x4*)((bf16_t*)p.out + orow * p.ldo + n) = u;
            *(bf16x4*)((bf16_t*)p.out2 + orow * p.ldo2 + n) =
                (bf16x4){f2bf(gelu_erf(bf2f(u[0]))), f2bf(gelu_erf(bf2f(u[1]))), f2bf(gelu_erf(bf2f(u[2]))), f2bf(gelu_erf(bf2f(u[3])))};
        } else if constexpr (EPI == VT_EPI_BF16_DGELU) {
            const bf16x4 uu = *(const bf16x4*)((const bf16_t*)p.aux + (int64_t)m * p.ldaux + n);
            *(bf16x4*)((bf16_t*)p.out + orow * p.ldo + n) =
                (bf16x4){f2bf(v[0] * gelu_erf_grad(bf2f(uu[0]))), f2bf(v[1] * gelu_erf_grad(bf2f(uu[1]))),
                         f2bf(v[2] * gelu_erf_grad(bf2f(uu[2]))), f2bf(v[3] * gelu_erf_grad(bf2f(uu[3])))};
        } else {  // VT_EPI_F32
            if (p.round_bf16) v = (f32x4){round_bf16(v[0]), round_bf16(v[1]), round_bf16(v[2]), round_bf16(v[3])};
            if (p.residual) v += *(const f32x4*)(p.residual + orow * p.ldr + n);   // (a streaming load of these 64-B line halves fetches every line twice)
            if (p.rowmod) v += *(const f32x4*)(p.rowmod + (int64_t)(m % p.rowmod_period) * p.N + n);
            if (p.out_scale != 0.f) v *= p.out_scale;
            *(f32x4*)((float*)p.out + orow * p.ldo + n) = v;   // 64-B pieces of a line per instruction: a streaming store would lose the L2's write combining (+45 %)
            if (p.out2) *(bf16x4*)((bf16_t*)p.out2 + orow * p.ldo2 + n) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
        }
        return;
    }
    // ragged tail of the matrix (N % 4 != 0 only): scalar
    for (int r = 0; r < 4 && n + r < p.N; ++r) {
        float v = acc4[r] + (p.bias ? p.bias[n + r] : 0.f);
        if constexpr (EPI == VT_EPI_BF16) {
            ((bf16_t*)p.out)[orow * p.ldo + n + r] = f2bf(v);
        } else if constexpr (EPI == VT_EPI_BF16_GELU) {
            const bf16_t u = f2bf(v);
            ((bf16_t*)p.out)[orow * p.ldo + n + r] = u;
            ((bf16_t*)p.out2)[orow * p.ldo2 + n + r] = f2bf(gelu_erf(bf2f(u)));
        } else if constexpr (EPI == VT_EPI_BF16_DGELU) {
            const float u = bf2f(((const bf16_t*)p.aux)[(int64_t)m * p.ldaux + n + r]);
            ((bf16_t*)p.out)[orow * p.ldo + n + r] = f2bf(v * gelu_erf_grad(u));
        } else {
            if (p.round_bf16) v = round_bf16(v);
            if (p.residual) v += p.residual[orow * p.ldr + n + r];
            if (p.rowmod) v += p.rowmod[(int64_t)(m % p.rowmod_period) * p.N + n + r];
            if (p.out_scale != 0.f) v *= p.out_scale;
            ((float*)p.out)[orow * p.ldo + n + r] = v;
            if (p.out2) ((bf16_t*)p.out2)[orow * p.ldo2 + n + r] = f2bf(v);
        }
    }
}
