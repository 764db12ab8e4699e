// (T,H,W) patchify / unpatchify for gfx950 -- HBM-bound, 16-32 B per lane per access.
//
// patchify  : video fp32 [B,C,T,H,W]  ->  patch rows bf16 [B*Nv, Kp], Kp = C*pt*p*p, column order
//             (c, dt, dy, dx) == the flattening of the Conv3d weight [D,C,pt,p,p] (models/embed.py:82),
//             token order (t,h,w) t-major (embed.py:112).  Also used on d(pred_frames) in backward.
// unpatchify: rows fp32 [B*Nv, Kp] in the SAME (c,dt,dy,dx) column order -> video fp32 [B,C,T,H,W].
//             The reference's head emits columns in (dt,dy,dx,c) order (channel last,
//             models/larp_tokenizer.py:452-453); the head weight rows are permuted once at pack time
//             so both directions share this coalesced layout.
// Threads are laid out over the VIDEO index space so the video side is fully coalesced (each lane
// moves 8 consecutive pixels of one image row); the patch-row side moves 16-B (bf16) / 32-B (fp32)
// pieces, 2+ lanes per contiguous run.
#include "vt_common.h"

namespace {

struct PatchGeom {
    int B, C, T, H, W, pt, p;
    int Th, Hh, Ww;  // token grid
    int Kp;
};

__device__ __forceinline__ void decode(const PatchGeom& g, int64_t idx, int& b, int& c, int& t, int& y, int& x8) {
    const int w8 = g.W >> 3;
    x8 = (int)(idx % w8) * 8;
    idx /= w8;
    y = (int)(idx % g.H);
    idx /= g.H;
    t = (int)(idx % g.T);
    idx /= g.T;
    c = (int)(idx % g.C);
    b = (int)(idx / g.C);
}

__device__ __forceinline__ int64_t patch_offset(const PatchGeom& g, int b, int c, int t, int y, int x) {
    const int tt = t / g.pt, dt = t % g.pt, hh = y / g.p, dy = y % g.p, ww = x / g.p, dx = x % g.p;
    const int64_t tok = ((int64_t)b * g.Th + tt) * g.Hh * g.Ww + (int64_t)hh * g.Ww + ww;
    const int col = ((c * g.pt + dt) * g.p + dy) * g.p + dx;
    return tok * g.Kp + col;
}

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ v, PatchGeom g, bf16_t* __restrict__ out, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int b, c, t, y, x;
        decode(g, idx, b, c, t, y, x);
        const float* src = v + ((((int64_t)b * g.C + c) * g.T + t) * g.H + y) * g.W + x;
        const f32x4 a0 = *(const f32x4*)src, a1 = *(const f32x4*)(src + 4);
        bf16x8 o = {f2bf(a0[0]), f2bf(a0[1]), f2bf(a0[2]), f2bf(a0[3]), f2bf(a1[0]), f2bf(a1[1]), f2bf(a1[2]), f2bf(a1[3])};
        *(bf16x8*)(out + patch_offset(g, b, c, t, y, x)) = o;
    }
}

__global__ __launch_bounds__(256) void unpatchify_kernel(const float* __restrict__ rows, PatchGeom g, float* __restrict__ v, int64_t total) {
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        int b, c, t, y, x;
        decode(g, idx, b, c, t, y, x);
        const float* src = rows + patch_offset(g, b, c, t, y, x);
        float* dst = v + ((((int64_t)b * g.C + c) * g.T + t) * g.H + y) * g.W + x;
        *(f32x4*)dst = *(const f32x4*)src;
        *(f32x4*)(dst + 4) = *(const f32x4*)(src + 4);
    }
}

int make_geom(PatchGeom& g, int B, int C, int T, int S, int pt, int p) {
    if (B <= 0 || C <= 0 || T <= 0 || S <= 0 || pt <= 0 || p <= 0) return 0;
    if (T % pt || S % p || p % 8) return 0;
    g = PatchGeom{B, C, T, S, S, pt, p, T / pt, S / p, S / p, C * pt * p * p};
    return 1;
}

}  // namespace

extern "C" int vt_patchify(const float* video, int32_t B, int32_t C, int32_t T, int32_t S, int32_t pt, int32_t p, void* rows_bf16,
                           vtStream stream) {
    PatchGeom g;
    VT_CHECK_ARG(video && rows_bf16, "vt_patchify: null pointer");
    VT_CHECK_ARG(make_geom(g, B, C, T, S, pt, p), "vt_patchify: need T%%pt==0, S%%p==0, p%%8==0 (T=%d S=%d pt=%d p=%d)", T, S, pt, p);
    const int64_t total = (int64_t)B * C * T * S * (S / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(patchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, video, g, (bf16_t*)rows_bf16, total);
    VT_CHECK_LAUNCH("vt_patchify");
    return VT_OK;
}

extern "C" int vt_unpatchify(const float* rows, int32_t B, int32_t C, int32_t T, int32_t S, int32_t pt, int32_t p, float* video,
                             vtStream stream) {
    PatchGeom g;
    VT_CHECK_ARG(video && rows, "vt_unpatchify: null pointer");
    VT_CHECK_ARG(make_geom(g, B, C, T, S, pt, p), "vt_unpatchify: need T%%pt==0, S%%p==0, p%%8==0 (T=%d S=%d pt=%d p=%d)", T, S, pt, p);
    const int64_t total = (int64_t)B * C * T * S * (S / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, rows, g, video, total);
    VT_CHECK_LAUNCH("vt_unpatchify");
    return VT_OK;
}
