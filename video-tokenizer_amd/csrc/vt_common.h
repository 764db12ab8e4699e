// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the LARP tokenizer path.
// Wave = 64 lanes everywhere; no other architecture is targeted.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/vt_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VT_LDS __attribute__((address_space(3)))
#define VT_GLB __attribute__((address_space(1)))

// ---- error plumbing (thread-local message, negative codes across the C ABI) ----
void vt_set_error(const char* fmt, ...);
#define VT_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            vt_set_error(__VA_ARGS__);          \
            return VT_ERR_INVALID;              \
        }                                       \
    } while (0)
#define VT_CHECK_LAUNCH(name)                                              \
    do {                                                                   \
        hipError_t e__ = hipGetLastError();                                \
        if (e__ != hipSuccess) {                                           \
            vt_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return VT_ERR_LAUNCH;                                          \
        }                                                                  \
    } while (0)

// ---- device helpers ----
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

// 16-byte async global -> LDS copy: LDS destination = wave-uniform base + lane*16.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VT_GLB void*)gsrc, (VT_LDS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ void glds4(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VT_GLB void*)gsrc, (VT_LDS void*)lds_wave_base, 4, 0, 0);
}

// The same 16-byte LDS-DMA issued through inline asm: hipcc then does not know an LDS-DMA is in flight and keeps
// emitting COUNTED lgkmcnt waits for ds_reads (with the builtin in flight it degrades every LDS wait to
// lgkmcnt(0)).  The caller must order the DMA itself: s_waitcnt vmcnt(N) + barrier before any ds_read of the data.
// lds_byte_addr must be wave-uniform (SGPR); M0 is saved/restored inside the statement.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
__device__ __forceinline__ void glds4_asm(const void* gsrc, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dword %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_byte_addr)
        : "memory");
}
// 16-B LDS-DMA with a WAVE-UNIFORM 64-bit base (SGPR pair) and a per-lane 32-bit byte offset.  A streaming kernel computes
// the lane offsets once; per tile it only advances the scalar base, so staging costs no per-lane address arithmetic
// (the 64-bit multiply-adds of glds16_asm were 16 % of the attention forward, tools ablation in DESIGN.md).
__device__ __forceinline__ void glds16_sv(const void* base_uniform, unsigned lane_byte_off, unsigned lds_byte_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(lane_byte_off), "s"(base_uniform), "s"(lds_byte_addr)
        : "memory");
}
// drain this wave's LDS-DMA before a barrier that publishes the staged tile (the compiler does not see asm DMAs)
__device__ __forceinline__ void dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr_of(const void* p) { return (unsigned)(unsigned long long)(VT_LDS const char*)p; }

// transposed LDS read: per 16-lane group a 4-row x 16-col block of 16-bit elements; lane i of the
// group receives column i (4 rows).  addr = this lane's 8-byte piece (row q = (i>>2), cols 4*(i&3)..).
__device__ __forceinline__ bf16x4 lds_read_tr16(const void* lds_addr) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VT_LDS s16x4*)lds_addr);
    return __builtin_bit_cast(bf16x4, v);
}

__device__ __forceinline__ bf16x8 cat4(bf16x4 a, bf16x4 b) {
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// Streaming (non-temporal) global accesses for data that is touched once per kernel -- GEMM outputs, the residual and
// pre-activation operands of an epilogue: they should not displace the A / B panels that 4-16 neighbouring tiles re-read
// from the XCD's 4 MB L2 (PMC: the GELU-epilogue GEMM fetched 7x its algorithmic bytes with ordinary stores).
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ void st_stream(bf16x4* p, bf16x4 v) { __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, v), (u32x2_t*)p); }
__device__ __forceinline__ void st_stream(f32x4* p, f32x4 v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ bf16x4 ld_stream(const bf16x4* p) { return __builtin_bit_cast(bf16x4, __builtin_nontemporal_load((const u32x2_t*)p)); }
__device__ __forceinline__ f32x4 ld_stream(const f32x4* p) { return __builtin_nontemporal_load(p); }

// the same for any 4 / 8 / 16-byte vector type (LayerNorm rows are 8- or 16-byte per lane depending on the width)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
template <typename T>
__device__ __forceinline__ void st_stream_any(T* p, const T& v) {
    if constexpr (sizeof(T) == 16) __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), (u32x4_t*)p);
    else if constexpr (sizeof(T) == 8) __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, v), (u32x2_t*)p);
    else __builtin_nontemporal_store(__builtin_bit_cast(unsigned, v), (unsigned*)p);
}
template <typename T>
__device__ __forceinline__ T ld_stream_any(const T* p) {
    if constexpr (sizeof(T) == 16) return __builtin_bit_cast(T, __builtin_nontemporal_load((const u32x4_t*)p));
    else if constexpr (sizeof(T) == 8) return __builtin_bit_cast(T, __builtin_nontemporal_load((const u32x2_t*)p));
    else return __builtin_bit_cast(T, __builtin_nontemporal_load((const unsigned*)p));
}

__device__ __forceinline__ float bf2f(bf16_t x) { return (float)x; }
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)x; }
__device__ __forceinline__ float round_bf16(float x) { return (float)((bf16_t)x); }

// Exact-erf GELU (timm Mlp act_layer=nn.GELU) without the branchy library erff: Abramowitz-Stegun 7.1.26,
// erfc(z) = (a1 t + .. + a5 t^5) e^{-z^2}, t = 1/(1 + p z), |error| <= 1.5e-7 on erf -- three orders below the bf16
// rounding applied to every GELU output/gradient here.  cdf is formed without cancellation on the negative side,
// and the same exp(-x^2/2) serves the pdf term of the derivative.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    e = __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);  // exp(-x^2/2)
    const float hq = 0.5f * p * t * e;                             // 0.5 * erfc(|x|/sqrt2)
    cdf = x >= 0.f ? 1.0f - hq : hq;
}
__device__ __forceinline__ float gelu_erf(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf, e;
    gelu_parts(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// bijective XCD-aware remap (8 XCDs, blocks dealt round-robin): gives each XCD one contiguous chunk
// of the tile list so neighbouring tiles share operand panels through that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Tile order of an NT GEMM launch whose tile list is dealt to the XCDs in contiguous chunks (xcd_remap): 0 = row-major list, W > 0 = column blocks
// of W tile columns, row-major inside a block, so that a chunk is a rectangle of tiles and the tiles in flight on one XCD touch the fewest
// distinct operand panels (rows + columns of the rectangle; vt_gemm192.hip "Tile order of a launch" has the measurements).  `in_flight` = tiles
// one XCD works on at a time.  W is the block width nearest sqrt(in_flight) that cuts the tile columns into equal blocks (the last may be narrower).
static inline int vt_auto_col_block(int tiles_n, int in_flight) {
    if (tiles_n < 2 || in_flight < 4) return 0;
    int target = 1;
    while ((target + 1) * (target + 1) <= in_flight) ++target;            // floor(sqrt(in_flight))
    if ((target + 1) * (target + 1) - in_flight < in_flight - target * target) ++target;
    int k = (tiles_n + target / 2) / target;                                // column blocks
    if (k < 1) k = 1;
    const int w = (tiles_n + k - 1) / k;
    return w >= tiles_n ? 0 : w;
}
// (tile row, tile column) of list entry `sid` under that order
__host__ __device__ __forceinline__ void vt_tile_of(int sid, int tiles_m, int tiles_n, int col_block, int& tm, int& tn) {
    tm = sid / tiles_n, tn = sid - tm * tiles_n;
    if (col_block > 0) {
        const int per = tiles_m * col_block, b = sid / per, r = sid - b * per;
        const int left = tiles_n - b * col_block, w = left < col_block ? left : col_block;
        tm = r / w, tn = b * col_block + (r - tm * w);
    }
}

// output row map: r -> (r / grp) * stride + off + (r % grp); grp == 0 means identity
struct RowMap {
    int grp;
    int64_t stride, off;
    __host__ __device__ __forceinline__ int64_t operator()(int64_t r) const {
        return grp ? (r / grp) * stride + off + (r % grp) : r;
    }
};

// ---- library-private (not part of the C ABI): deferred partial-sum reductions of the tokenizer engine ----------------------
// The backward of a block leaves three sets of per-slab partial sums (fc1 bias gradient out of the gelu' epilogue, the two
// LayerNorm backward kernels' dgamma / dbeta / column sums).  Reducing each with its own launch costs a ~4.8 us kernel 82 times
// per step; the engine queues them and reduces the queue of 4 blocks in ONE launch next to the grouped weight-gradient GEMM.
struct vtReduceItem {
    const float* partial;   // [nslab][slab_stride]
    int nslab, width, nout, lanes;   // lanes = 8 or 32 slab lanes per column: the summation order of the stand-alone reducers
    int64_t slab_stride;
    float* o[3];            // o[w][c] = sum_s partial[s * slab_stride + w * width + c]
};
#define VT_REDUCE_MAX_GROUP 16
int vt_reduce_grouped(const vtReduceItem* items, int n, vtStream stream);
// LayerNorm backward without its reduction: partial sums go to `part` ([grid][3 * dim]); *nslab = grid
int vt_layernorm_bwd_partials(const void* dy_bf16, const float* x, vtRowMap xmap, const float* gamma, const float* mean, const float* rstd,
                              const float* dres, int64_t rows, int32_t dim, float* dx, void* dx_bf16, float* part, int* nslab, vtStream stream);

#endif
