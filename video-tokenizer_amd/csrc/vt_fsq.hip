// Finite scalar quantizer (FSQ) for gfx950: the regulariser of the reference's TiTok-style autoencoders
// (/root/reference/models/model_new/quantizer/fsq.py:54-131; instantiated with levels [8,8,8,5,5,5] or
// [8,8,8,8,5,5,5,5], models/model_new/autoencoder.py:59,140,640).  The reference runs ~12 elementwise torch ops plus a
// reduction over an [N, d<=8] tensor; here the whole quantizer is one launch per direction, one lane per token row.
//
//   bounded   = tanh(z + shift) * half_l - offset        fsq.py:76-81
//   quantized = rint(bounded)  (straight-through)        fsq.py:47-50,83-88
//   codes     = quantized / (levels // 2)
//   indices   = int32( sum_c (codes_c * hw_c + hw_c) * basis_c )      fsq.py:90-92,103-107
//
// All arithmetic is fp32 in the reference's operation order (the forward runs with autocast disabled, :119-131), so
// indices are bit-exact against oracle/fsq_oracle.c.  tanh is evaluated in double and rounded -- N*d is a few tens
// of thousands of elements, the fp64 rate is irrelevant next to the launch itself -- which gives the correctly
// rounded fp32 tanh, the same value the oracle computes with libm.
#include "vt_common.h"

#include <cmath>

#define FSQ_MAX_D 16

namespace {
#pragma clang fp contract(off)

struct FsqConsts {
    int d;
    int levels[FSQ_MAX_D];
    int basis[FSQ_MAX_D];
    float half_l[FSQ_MAX_D], offset[FSQ_MAX_D], shift[FSQ_MAX_D], half_width[FSQ_MAX_D];
};

// host side of fsq.py:62-73,78-80; identical to oracle/fsq_oracle.c:fsq_constants
bool make_consts(const int32_t* levels, int d, FsqConsts& k) {
    int64_t b = 1;
    k.d = d;
    for (int c = 0; c < d; ++c) {
        if (levels[c] < 2) return false;
        k.levels[c] = levels[c];
        k.half_l[c] = (float)(levels[c] - 1) * (float)(1.0 + 1e-3) / 2.0f;
        k.offset[c] = (levels[c] % 2 == 0) ? 0.5f : 0.0f;
        k.shift[c] = (float)atanh((double)(k.offset[c] / k.half_l[c]));
        k.half_width[c] = (float)(levels[c] / 2);
        k.basis[c] = (int)b;
        b *= levels[c];
        if (b > (1 << 24)) return false;  // the reference sums level indices in fp32: exact only below 2^24
    }
    return true;
}

template <typename T> __device__ __forceinline__ float ld(const T* p, int64_t i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void st(T* p, int64_t i, float v) { p[i] = (T)v; }

__device__ __forceinline__ float tanh_rn(float x) { return (float)tanh((double)x); }

template <typename T>
__global__ __launch_bounds__(256) void fsq_fwd_kernel(const T* __restrict__ z, int64_t N, FsqConsts k, T* __restrict__ codes,
                                                       int32_t* __restrict__ indices) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float acc = 0.0f;
    for (int c = 0; c < k.d; ++c) {
        const float t = tanh_rn(ld(z, n * k.d + c) + k.shift[c]);
        const float bounded = t * k.half_l[c] - k.offset[c];
        const float code = __fdiv_rn(rintf(bounded), k.half_width[c]);
        st(codes, n * k.d + c, code);
        acc = acc + (code * k.half_width[c] + k.half_width[c]) * (float)k.basis[c];
    }
    if (indices) indices[n] = (int32_t)acc;
}

template <typename T>
__global__ __launch_bounds__(256) void fsq_bwd_kernel(const T* __restrict__ z, const T* __restrict__ dcodes, int64_t total, FsqConsts k,
                                                       T* __restrict__ dz) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % k.d);
    const float t = tanh_rn(ld(z, i) + k.shift[c]);
    const float g = __fdiv_rn(ld(dcodes, i), k.half_width[c]);
    st(dz, i, (g * k.half_l[c]) * (1.0f - t * t));
}

template <typename T>
__global__ __launch_bounds__(256) void fsq_i2c_kernel(const int32_t* __restrict__ indices, int64_t total, FsqConsts k, T* __restrict__ codes) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % k.d);
    const int lvl = (indices[i / k.d] / k.basis[c]) % k.levels[c];
    st(codes, i, __fdiv_rn((float)lvl - k.half_width[c], k.half_width[c]));
}

int check(const char* name, int64_t N, int d, const int32_t* levels_host, FsqConsts& k) {
    VT_CHECK_ARG(N > 0 && d >= 1 && d <= FSQ_MAX_D && levels_host, "%s: need N > 0, 1 <= d <= %d and a host levels array", name, FSQ_MAX_D);
    VT_CHECK_ARG(make_consts(levels_host, d, k), "%s: every level must be >= 2 and prod(levels) <= 2^24", name);
    return VT_OK;
}
}  // namespace

extern "C" int vt_fsq_codebook_size(const int32_t* levels_host, int32_t d, int64_t* size) {
    FsqConsts k;
    VT_CHECK_ARG(levels_host && size && d >= 1 && d <= FSQ_MAX_D && make_consts(levels_host, d, k), "vt_fsq_codebook_size: bad levels");
    *size = (int64_t)k.basis[d - 1] * levels_host[d - 1];
    return VT_OK;
}

extern "C" int vt_fsq_forward(const void* z, int32_t is_bf16, int64_t N, int32_t d, const int32_t* levels_host, void* codes, int32_t* indices,
                              vtStream stream) {
    FsqConsts k;
    if (int rc = check("vt_fsq_forward", N, d, levels_host, k)) return rc;
    VT_CHECK_ARG(z && codes, "vt_fsq_forward: null z/codes");
    const dim3 grid((unsigned)((N + 255) / 256));
    if (is_bf16)
        hipLaunchKernelGGL(fsq_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, N, k, (bf16_t*)codes, indices);
    else
        hipLaunchKernelGGL(fsq_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, N, k, (float*)codes, indices);
    VT_CHECK_LAUNCH("vt_fsq_forward");
    return VT_OK;
}

extern "C" int vt_fsq_backward(const void* z, const void* dcodes, int32_t is_bf16, int64_t N, int32_t d, const int32_t* levels_host, void* dz,
                               vtStream stream) {
    FsqConsts k;
    if (int rc = check("vt_fsq_backward", N, d, levels_host, k)) return rc;
    VT_CHECK_ARG(z && dcodes && dz, "vt_fsq_backward: null pointer");
    const int64_t total = N * d;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (is_bf16)
        hipLaunchKernelGGL(fsq_bwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)z, (const bf16_t*)dcodes, total, k, (bf16_t*)dz);
    else
        hipLaunchKernelGGL(fsq_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)z, (const float*)dcodes, total, k, (float*)dz);
    VT_CHECK_LAUNCH("vt_fsq_backward");
    return VT_OK;
}

extern "C" int vt_fsq_indices_to_codes(const int32_t* indices, int64_t N, int32_t d, const int32_t* levels_host, void* codes, int32_t is_bf16,
                                       vtStream stream) {
    FsqConsts k;
    if (int rc = check("vt_fsq_indices_to_codes", N, d, levels_host, k)) return rc;
    VT_CHECK_ARG(indices && codes, "vt_fsq_indices_to_codes: null pointer");
    const int64_t total = N * d;
    const dim3 grid((unsigned)((total + 255) / 256));
    if (is_bf16)
        hipLaunchKernelGGL(fsq_i2c_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, indices, total, k, (bf16_t*)codes);
    else
        hipLaunchKernelGGL(fsq_i2c_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, indices, total, k, (float*)codes);
    VT_CHECK_LAUNCH("vt_fsq_indices_to_codes");
    return VT_OK;
}
