// Tile-level helpers shared by the attention kernels (vt_attention.hip, vt_attention_bwd.hip): the [64][HD] bf16 LDS tile image,
// its swizzle, LDS-DMA staging, the row / transposed MFMA fragments and the accumulator-as-operand packing.  See the header
// comment of vt_attention.hip for the layout and the k-order convention.
#pragma once
#include "vt_common.h"

namespace {


template <int HD>
struct AG {
    static constexpr int ROWB = HD * 2;       // bytes per tile row
    static constexpr int TILE = 64 * ROWB;    // bytes per [64][HD] bf16 tile
    static constexpr int KS = HD / 16;        // k-steps of a product that contracts over head_dim
    static constexpr int DT = HD / 32;        // 32-wide output tiles along head_dim
    static constexpr int CH = HD / 8;         // 16-B chunks per row
};

template <int HD>
__device__ __forceinline__ int fsw(int row) {
    return HD == 64 ? ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) : ((row >> 2) & 3);
}

// LDS-DMA through inline asm (see glds16_asm): callers drain with dma_drain() before the publishing barrier
template <int HD>
__device__ __forceinline__ void stage64(const bf16_t* __restrict__ src, int64_t rs, int row0, int L, unsigned lds, int tid, int wave) {
    constexpr int CH = AG<HD>::CH;
#pragma unroll
    for (int i = 0; i < CH / 4; ++i) {
        const int slot = i * 256 + tid;
        const int row = slot / CH;
        const int lc = (slot % CH) ^ fsw<HD>(row);
        int gr = row0 + row;
        gr = gr < L ? gr : L - 1;
        glds16_asm(src + (int64_t)gr * rs + lc * 8, lds + (i * 256 + wave * 64) * 16);
    }
}

// Full (unclamped) tiles of the steady state: lane offsets inside a [64][HD] tile are loop invariants (stage_offsets), the
// tile's first row is a scalar base pointer.
template <int HD>
__device__ __forceinline__ void stage_offsets(int64_t rs, int tid, unsigned (&off)[AG<HD>::CH / 4]) {
    constexpr int CH = AG<HD>::CH;
#pragma unroll
    for (int i = 0; i < CH / 4; ++i) {
        const int slot = i * 256 + tid;
        const int row = slot / CH;
        const int lc = (slot % CH) ^ fsw<HD>(row);
        off[i] = (unsigned)((row * rs + lc * 8) * 2);
    }
}
template <int HD>
__device__ __forceinline__ void stage64_full(const bf16_t* tile_row0, const unsigned (&off)[AG<HD>::CH / 4], unsigned lds, int wave) {
#pragma unroll
    for (int i = 0; i < AG<HD>::CH / 4; ++i) glds16_sv(tile_row0, off[i], lds + (i * 256 + wave * 64) * 16);
}

// A operand of a 32x32x16 MFMA from tile rows r0..r0+31, k-step s (16 columns)
template <int HD>
__device__ __forceinline__ bf16x8 rowfrag(const char* lds, int r0, int s, int lane) {
    const int row = r0 + (lane & 31);
    const int lc = 2 * s + (lane >> 5);
    return *(const bf16x8*)(lds + row * AG<HD>::ROWB + ((lc ^ fsw<HD>(row)) << 4));
}

// A operand [i = column c0 + (lane&31)][k = tile rows], k order matched to an accumulator used as B:
// element j  <->  tile row rbase + 16*sp + 8*(j>>2) + 4*(lane>>5) + (j&3)
template <int HD>
__device__ __forceinline__ bf16x8 trfrag(const char* lds, int rbase, int sp, int c0, int lane) {
    const int g = lane >> 4, lam = lane & 15;
    const int r0 = rbase + 16 * sp + 4 * (g >> 1) + (lam >> 2);
    const int r1 = r0 + 8;
    const int cb = c0 + 16 * (g & 1);
    const int lc = (cb >> 3) + ((lam & 3) >> 1);
    const int bo = (lam & 1) << 3;
    const bf16x4 lo = lds_read_tr16(lds + r0 * AG<HD>::ROWB + ((lc ^ fsw<HD>(r0)) << 4) + bo);
    const bf16x4 hi = lds_read_tr16(lds + r1 * AG<HD>::ROWB + ((lc ^ fsw<HD>(r1)) << 4) + bo);
    return cat4(lo, hi);
}

// Round 4: the same two fragments from per-lane byte offsets computed ONCE per kernel.  Everything that depends on the buffer, the tile
// (K / V / Q / dO image), the 32-row half or the 16-row k-step of a transposed read is a multiple of 16 rows -- the swizzle has period
// 16 rows -- so it is a compile-time constant the caller adds to the image pointer and hipcc folds into the instruction's offset field.
// A wave then carries KS + 2 * DT address registers (8 at head_dim 64) and spends no vector instruction on LDS addressing inside
// the tile loop (ISA audit of the round-3 kernels, profiles/r04_attention_isa_audit.txt: 1.0-1.6 address instructions per MFMA).
template <int HD>
struct TileAddr {
    unsigned k[AG<HD>::KS];        // row fragment of k-step s: row lane & 31
    unsigned v[AG<HD>::DT][2];     // transposed fragment of column block dt: rows 4 (g >> 1) + (lam >> 2) and + 8
};
// `base` = LDS byte address of the first tile image (lds_addr_of(smem)): the registers hold ABSOLUTE LDS addresses, so a read is
// register + immediate and nothing else (with offsets relative to `smem` hipcc re-added the symbol's address at every use)
template <int HD>
__device__ __forceinline__ TileAddr<HD> tile_addr(int lane, unsigned base) {
    TileAddr<HD> a;
    const int row = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int s = 0; s < AG<HD>::KS; ++s) a.k[s] = base + row * AG<HD>::ROWB + (((2 * s + hf) ^ fsw<HD>(row)) << 4);
    const int g = lane >> 4, lam = lane & 15;
    const int r0 = 4 * (g >> 1) + (lam >> 2), r1 = r0 + 8;
    const int bo = (lam & 1) << 3;
#pragma unroll
    for (int dt = 0; dt < AG<HD>::DT; ++dt) {
        const int lc = ((dt * 32 + 16 * (g & 1)) >> 3) + ((lam & 3) >> 1);
        a.v[dt][0] = base + r0 * AG<HD>::ROWB + ((lc ^ fsw<HD>(r0)) << 4) + bo;
        a.v[dt][1] = base + r1 * AG<HD>::ROWB + ((lc ^ fsw<HD>(r1)) << 4) + bo;
    }
    // opaque from here on: under register pressure hipcc otherwise re-derives these from the lane id inside the tile loop
    // (32 v_xor per two tiles in the dK/dV kernel), which is exactly the address arithmetic this struct exists to remove
#pragma unroll
    for (int s = 0; s < AG<HD>::KS; ++s) asm volatile("" : "+v"(a.k[s]));
#pragma unroll
    for (int dt = 0; dt < AG<HD>::DT; ++dt) {
        asm volatile("" : "+v"(a.v[dt][0]));
        asm volatile("" : "+v"(a.v[dt][1]));
    }
    return a;
}
// img: byte offset of the tile image behind `base` (a compile-time constant in the unrolled loops); r0: first row of the 32-row
// block (0 or 32).  (r0, s, sp, dt are compile-time constants at every call site once the tile bodies are unrolled.)
__device__ __forceinline__ bf16x8 lds_ld128(unsigned addr) { return *(const VT_LDS bf16x8*)(uintptr_t)addr; }
__device__ __forceinline__ f32x4 lds_ld128f(unsigned addr) { return *(const VT_LDS f32x4*)(uintptr_t)addr; }
__device__ __forceinline__ bf16x4 lds_tr16(unsigned addr) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((VT_LDS s16x4*)(uintptr_t)addr));
}
template <int HD>
__device__ __forceinline__ bf16x8 rowfrag_a(unsigned img, const TileAddr<HD>& a, int r0, int s) {
    return lds_ld128(a.k[s] + img + r0 * AG<HD>::ROWB);
}
template <int HD>
__device__ __forceinline__ bf16x8 trfrag_a(unsigned img, const TileAddr<HD>& a, int r0, int sp, int dt) {
    const unsigned cst = img + (r0 + 16 * sp) * AG<HD>::ROWB;
    return cat4(lds_tr16(a.v[dt][0] + cst), lds_tr16(a.v[dt][1] + cst));
}

typedef __attribute__((ext_vector_type(2))) float f32x2;

// own 32 rows as B-operand fragments straight from global: f[s] = X[row0 + (lane&31)][16s + 8*(lane>>5) ..+7]
template <int KS>
__device__ __forceinline__ void load_own(const bf16_t* __restrict__ src, int64_t rs, int row0, int L, int lane, bf16x8 (&f)[KS]) {
    int r = row0 + (lane & 31);
    r = r < L ? r : L - 1;
    const bf16_t* p = src + (int64_t)r * rs + 8 * (lane >> 5);
#pragma unroll
    for (int s = 0; s < KS; ++s) f[s] = *(const bf16x8*)(p + 16 * s);
}

// Own-row operands arrive by ordinary global loads that hipcc counts; the tile loops stage by inline-asm LDS-DMA that it
// does not.  Left alone, hipcc places the wait for the own-row loads at their first use INSIDE the loop as vmcnt(3..0) --
// which at run time drains the next tile's just-issued DMA at the top of every iteration (the prefetch never overlapped
// the tile's compute).  Making the registers opaque here puts that wait in front of the loop, once.
template <int N>
__device__ __forceinline__ void pin_loaded(bf16x8 (&f)[N]) {
#pragma unroll
    for (int s = 0; s < N; ++s) asm volatile("" : "+v"(f[s]));
}
__device__ __forceinline__ void pin_loaded(float& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int sp) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = f2bf(a[8 * sp + j]);
    return r;
}

__device__ __forceinline__ int reg_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// store acc^T[d][own_row] tiles (DT x f32x16) as bf16 into dst[own_row][d].  After the 32x32 MFMA chain a row is split over the two
// half-waves (lane i: columns 8k .. 8k+3, lane i + 32: columns 8k+4 .. 8k+7), so the natural store is 8 bytes per lane and an
// instruction touches 32 lines with 16 bytes each: the tail is store-ISSUE-bound (guide T21).  v_permlane32_swap exchanges the
// upper half-wave's group k with the lower one's group k + 1: lanes 0-31 then hold columns 8k .. 8k+7 and lanes 32-63 columns
// 8k+8 .. 8k+15 of their row -- ONE 16-byte store per pair of groups, same bytes at the same addresses, half the instructions.
// (every lane takes part in the swaps; `ok` only masks the stores: a row's two lanes are valid or invalid together)
#ifndef VT_ATTN_STORE8
#define VT_ATTN_STORE8 0     // A/B diagnostic: 1 = the 8-byte stores of rounds 1-5
#endif
template <int DT>
__device__ __forceinline__ void store_own(const f32x16 (&acc)[DT], float mul, bf16_t* __restrict__ dst, int64_t rs, int row, bool ok, int half) {
    if constexpr (VT_ATTN_STORE8) {
        if (!ok) return;
        bf16_t* p = dst + (int64_t)row * rs;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = f2bf(acc[dt][4 * g4 + e] * mul);
                *(bf16x4*)(p + dt * 32 + 8 * g4 + 4 * half) = v;
            }
    } else {
        bf16_t* p = dst + (int64_t)row * rs + 8 * half;      // the upper half-wave writes the next 16 bytes of the row
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int k = 0; k < 4; k += 2) {
                bf16x4 va, vb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    va[e] = f2bf(acc[dt][4 * k + e] * mul);
                    vb[e] = f2bf(acc[dt][4 * (k + 1) + e] * mul);
                }
                u32x2_t a = __builtin_bit_cast(u32x2_t, va), b = __builtin_bit_cast(u32x2_t, vb);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a[0], b[0], false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a[1], b[1], false, false);
                const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
                if (ok) *(u32x4_t*)(p + dt * 32 + 8 * k) = w;
            }
    }
}

}  // namespace
