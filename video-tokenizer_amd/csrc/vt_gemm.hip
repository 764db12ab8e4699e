// bf16 MFMA GEMMs for gfx950.
//
//  gemm_nt_kernel : C[M,N] = A[M,K] . B[N,K]^T   (both operands K-contiguous)  + fused epilogues
//  gemm_tn_kernel : C[P,Q] = A[M,P]^T . B[M,Q]   (contraction over the ROW index of both operands:
//                   fragments come from LDS through the transposed read ds_read_b64_tr_b16)
//
// Structure of both: 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave,
// 4x4 accumulators of v_mfma_f32_16x16x32_bf16), BK = 64, operands staged global -> LDS with
// 16-byte global_load_lds into a lane-linear image; the XOR swizzle that makes the fragment reads
// bank-conflict-free is applied to the per-lane SOURCE address and to the read address (never to the
// LDS destination).  Two LDS buffers: the loads of K-tile t+1 are in flight while tile t is
// multiplied; one barrier per K-tile.  64 KiB LDS => 2 workgroups per CU.
//
// The MFMA is issued as D' = B_frag x A_frag so that one lane ends up with FOUR CONSECUTIVE OUTPUT
// COLUMNS of one output row (16 B of fp32 / 8 B of bf16 per store) instead of four rows.
#include <cstdlib>
#include "vt_common.h"
#include "vt_gemm_epilogue.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_B = BM * BK * 2;  // 16 KiB per operand tile

// ------------------------------------------------------------------------------------------------
// NT
// ------------------------------------------------------------------------------------------------
struct NTArgs {
    vtGemmNT p;
    int tiles_m, tiles_n;
    int col_block;      // tile order (vt_tile_of): 0 = row-major list, W = column blocks of W tile columns
    int split;          // workgroups per output tile (split K), 1 = none
    float* part;        // split > 1: fp32 partial tiles, [tile][split][16 acc groups][256 threads] f32x4 (what each lane holds)
    unsigned* ctr;      // split > 1: one arrival counter per tile, zero before and after every launch
};

// split-K hand-off (Guideline 16 of the CDNA guide; the recipe vt_attention_bwd.hip measured): partial sums leave through
// write-through stores, the arrival counter is a device-scope atomic, and the last workgroup to arrive reads all the partials
// back past its XCD's L2 (sc1) -- in split order, its own included, so the sum does not depend on which workgroup came last.
__device__ __forceinline__ void st_part(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// [128 rows][64 k] bf16 tile, 128-B rows, 16-B chunks; physical chunk = logical ^ ((row>>1)&7)
__device__ __forceinline__ void stage_nt(const bf16_t* __restrict__ g, int64_t ld, int row0, int nrows, int k0,
                                         char* lds, int tid, int wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = i * 256 + tid;
        const int row = slot >> 3;
        const int lc = (slot & 7) ^ ((row >> 1) & 7);
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        glds16(g + (int64_t)gr * ld + k0 + lc * 8, lds + (i * 256 + wave * 64) * 16);
    }
}

// 16-byte LDS read the compiler does not track: the caller counts lgkmcnt itself (gemm_nt_kernel<., 4>)
template <int OFF>
__device__ __forceinline__ void lds_read16(bf16x8& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}

__device__ __forceinline__ bf16x8 frag_nt(const char* lds, int row, int lchunk) {
    return *(const bf16x8*)(lds + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4));
}

// NST = 2: the ring described in the header (two workgroups per CU: one's loads land under the other's MFMAs).
// NST = 4: the same tile, same MFMA order (bit-identical results) behind a 4-deep ring filled by inline-asm LDS-DMA, counted
//          s_waitcnt vmcnt(16) + raw s_barrier per K-tile (two K-tiles stay in flight across the barrier), fragment reads of one
//          half K-tile under the MFMAs of the other -- for launches with at
//          most one workgroup per CU (M = 1536 or 3072: one or two clips per GPU), where nothing else hides the ~1 us a K-tile's
//          loads take and the 2-deep ring ran one K-tile per load latency (fc2 forward at one clip: 48 K-tiles = 50 us on 72 CUs).
template <int EPI, int NST>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const NTArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const vtGemmNT& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = a.tiles_m * a.tiles_n;
    // split K: neighbouring workgroups (one XCD) take neighbouring tiles of the SAME K range, so they share operand panels in L2
    const int sid_all = xcd_remap(blockIdx.x, nwg * a.split);
    const int kz = sid_all / nwg, sid = sid_all - kz * nwg;
    int tm, tn;
    vt_tile_of(sid, a.tiles_m, a.tiles_n, a.col_block, tm, tn);
    const int m0 = __builtin_amdgcn_readfirstlane(tm) * BM, n0 = __builtin_amdgcn_readfirstlane(tn) * BN;
    const int nt_all = p.K / BK;
    const int t_first = (int)((long)kz * nt_all / a.split), nt = (int)((long)(kz + 1) * nt_all / a.split) - t_first;

    const bf16_t* A = (const bf16_t*)p.A + (int64_t)t_first * BK;
    const bf16_t* B = (const bf16_t*)p.B + (int64_t)t_first * BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS: [buffer 0: A | B][buffer 1: A | B]

    const int fr = lane & 15, fq = lane >> 4;
    // one K-tile of MFMAs out of the stage at `la` (A image) / `la + TILE_B` (B image)
    auto multiply = [&](const char* la) {
        const char* lb = la + TILE_B;
        bf16x8 af[2][4], bfv[2][4];  // all 16 fragment reads first, MFMAs behind counted lgkmcnt waits
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bfv[kk][j] = frag_nt(lb, wc * 64 + j * 16 + fr, kk * 4 + fq);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[kk][i] = frag_nt(la, wr * 64 + i * 16 + fr, kk * 4 + fq);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (NST == 2) {
        stage_nt(A, p.lda, m0, p.M, 0, smem, tid, wave);
        stage_nt(B, p.ldb, n0, p.N, 0, smem + TILE_B, tid, wave);
        __syncthreads();  // hipcc drains the LDS-DMA (vmcnt(0)) ahead of the barrier
        for (int t = 0; t < nt; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt) {
                stage_nt(A, p.lda, m0, p.M, (t + 1) * BK, smem + (cur ^ 1) * 2 * TILE_B, tid, wave);
                stage_nt(B, p.ldb, n0, p.N, (t + 1) * BK, smem + (cur ^ 1) * 2 * TILE_B + TILE_B, tid, wave);
            }
            multiply(smem + cur * 2 * TILE_B);
            __syncthreads();
        }
    } else {
        constexpr int STAGE = 2 * TILE_B;
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
        unsigned offa[4], offb[4];      // loop-invariant lane offsets (bytes) inside A / B; the K-tile moves the scalar base
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int slot = i * 256 + tid;
            const int row = slot >> 3;
            const int lc = (slot & 7) ^ ((row >> 1) & 7);
            int ra = m0 + row, rb = n0 + row;
            ra = ra < p.M ? ra : p.M - 1;
            rb = rb < p.N ? rb : p.N - 1;
            // relative to the tile's first row (<= 127 rows x ld): the 32-bit lane offset cannot wrap however large M x lda is
            // (absolute rows wrapped beyond 4 GiB of operand; the tile's base moves into the scalar pointer instead)
            offa[i] = (unsigned)(((int64_t)(ra - m0) * p.lda + lc * 8) * 2);
            offb[i] = (unsigned)(((int64_t)(rb - n0) * p.ldb + lc * 8) * 2);
        }
        const bf16_t* Atile = A + (int64_t)m0 * p.lda;
        const bf16_t* Btile = B + (int64_t)n0 * p.ldb;
        auto issue = [&](int t, int slot) {
            const bf16_t* ak = Atile + (int64_t)t * BK;
            const bf16_t* bk = Btile + (int64_t)t * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_sv(ak, offa[i], sbase + slot * STAGE + (i * 256 + wave * 64) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16_sv(bk, offb[i], sbase + slot * STAGE + TILE_B + (i * 256 + wave * 64) * 16);
        };
#pragma unroll
        for (int s_ = 0; s_ < NST; ++s_)
            if (s_ < nt) issue(s_, s_);
        if (nt >= NST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 1)) : "memory");   // K-tile 0 landed, the next NST-1 may fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // With one workgroup per CU nothing else covers a wave's fragment reads, and read-all-then-multiply ran them back to back with
        // its MFMAs (0.68 us per K-tile, three times the MFMA time).  Software pipeline over half K-tiles instead: the reads of one half
        // (8 x ds_read_b128) fly under the 16 MFMAs of the other, across the barrier too -- the second half of K-tile t is multiplied
        // AFTER the barrier, out of registers, while the first fragments of K-tile t+1 are being read.  Same MFMA order per accumulator
        // (half 0 then half 1 of every K-tile): results stay bit-identical to the 2-deep ring.
        // The fragment reads are inline asm with hand-counted s_waitcnt lgkmcnt: hipcc's own counter tracking gives up on loads that
        // are still pending across the loop's back edge (it emitted lgkmcnt(0) in front of the first MFMA, i.e. no overlap at all).
        bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
        unsigned aoff[2], boff[2];                          // lane offsets of the first A / B fragment of each half inside a stage
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            aoff[kk] = (unsigned)((wr * 64 + fr) * 128 + (((kk * 4 + fq) ^ ((fr >> 1) & 7)) << 4));
            boff[kk] = (unsigned)(TILE_B + (wc * 64 + fr) * 128 + (((kk * 4 + fq) ^ ((fr >> 1) & 7)) << 4));
        }
        auto frags = [&](int slot_, int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
            const unsigned sb = sbase + slot_ * STAGE;
            const unsigned ab = sb + aoff[kk], bb = sb + boff[kk];
            lds_read16<0>(fb[0], bb); lds_read16<2048>(fb[1], bb); lds_read16<4096>(fb[2], bb); lds_read16<6144>(fb[3], bb);
            lds_read16<0>(fa[0], ab); lds_read16<2048>(fa[1], ab); lds_read16<4096>(fa[2], ab); lds_read16<6144>(fa[3], ab);
        };
        auto mma = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[4]) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        };
        int slot = 0;                                       // ring position of the K-tile being multiplied
        frags(0, 0, fa0, fb0);
        for (int t = 0; t < nt; ++t) {
            frags(slot, 1, fa1, fb1);
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                // first half in registers, second half in flight
            __builtin_amdgcn_sched_barrier(0);
            mma(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + NST - 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 2)) : "memory");   // K-tile t+1 landed, t+2 and t+3 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                // this wave holds both halves of K-tile t
            __builtin_amdgcn_s_barrier();
            if (t + NST < nt) issue(t + NST, slot);         // every wave has K-tile t in registers: its slot takes K-tile t+4
            slot = slot + 1 == NST ? 0 : slot + 1;
            frags(slot, 0, fa0, fb0);                       // after the last K-tile: a slot nobody uses, read and dropped
            __builtin_amdgcn_sched_barrier(0);
            mma(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the dropped read: nothing of it may land on the epilogue's registers
    }

    if (a.split > 1) {
        float* mine = a.part + ((size_t)sid * a.split + kz) * (BM * BN);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) st_part(mine + ((i * 4 + j) * 256 + tid) * 4, acc[i][j]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // every partial of this workgroup is past the L2
        if (tid == 0) *(unsigned*)smem = __hip_atomic_fetch_add(a.ctr + sid, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned arrived = *(const unsigned*)smem;
        __syncthreads();                                   // the epilogue reuses smem
        if ((int)arrived != a.split - 1) return;
        if (tid == 0) __hip_atomic_store(a.ctr + sid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.part + (size_t)sid * a.split * (BM * BN)), 0, a.split * BM * BN * 4, 0x00020000);
        // a read that goes past the L2 takes ~2 us whatever its size: the first three partials are requested together (192 registers
        // while the accumulators are dead), later ones two at a time; the additions stay in split order
        auto fetch = [&](int z, f32x4 (&dst)[16]) {
#pragma unroll
            for (int g = 0; g < 16; ++g)
                dst[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (g * 256 + tid) * 16, z * (BM * BN * 4), 16));
        };
        f32x4 p0[16], p1[16], p2[16];
        fetch(0, p0);
        fetch(1, p1);
        if (a.split > 2) fetch(2, p2);
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            f32x4 v = p0[g] + p1[g];
            if (a.split > 2) v += p2[g];
            acc[g >> 2][g & 3] = v;
        }
        for (int z = 3; z < a.split; z += 2) {
            const bool two = z + 1 < a.split;
            fetch(z, p0);
            if (two) fetch(z + 1, p1);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                acc[g >> 2][g & 3] += p0[g];
                if (two) acc[g >> 2][g & 3] += p1[g];
            }
        }
    }

    // ---- epilogue: acc[i][j][r] = C[m0 + wr*64 + i*16 + fr][n0 + wc*64 + j*16 + fq*4 + r]
    const RowMap omap{p.omap.grp, p.omap.stride, p.omap.off};
    if ((p.N & 3) == 0) {
        // Outputs leave through LDS, like the 192x192 kernel's: written in the MFMA layout (a lane owns 4 consecutive columns
        // of 16 different rows), read back row-major so that a store instruction covers whole 256- / 512-byte row pieces.
        // The loop's last barrier has retired every read of the operand ring.
        f32x4 b4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + fq * 4;
            b4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (EPI != VT_EPI_F32) {
            constexpr int STRIDE = BN * 2 + 8;   // bytes; +8: the 16 rows of a ds_write_b64 group fall on different bank pairs
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 v = acc[i][j] + b4[j];
                    *(bf16x4*)(smem + (wr * 64 + i * 16 + fr) * STRIDE + (wc * 16 + j * 4 + fq) * 8) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            __syncthreads();
#pragma unroll 2
            for (int u = 0; u < BM * (BN / 4) / 256; ++u) {
                const int slot = u * 256 + tid;
                const int row = slot >> 5, c = slot & 31;
                const int m = m0 + row, n = n0 + c * 4;
                if (m >= p.M || n >= p.N) continue;
                const bf16x4 h = *(const bf16x4*)(smem + row * STRIDE + c * 8);
                bf16_t* o = (bf16_t*)p.out + (int64_t)m * p.ldo + n;
                if constexpr (EPI == VT_EPI_BF16) {
                    st_stream((bf16x4*)o, h);
                } else if constexpr (EPI == VT_EPI_BF16_GELU) {
                    st_stream((bf16x4*)o, h);
                    st_stream((bf16x4*)((bf16_t*)p.out2 + (int64_t)m * p.ldo2 + n),
                              (bf16x4){f2bf(gelu_erf(bf2f(h[0]))), f2bf(gelu_erf(bf2f(h[1]))), f2bf(gelu_erf(bf2f(h[2]))), f2bf(gelu_erf(bf2f(h[3])))});
                } else {
                    const bf16x4 uu = ld_stream((const bf16x4*)((const bf16_t*)p.aux + (int64_t)m * p.ldaux + n));
                    st_stream((bf16x4*)o, (bf16x4){f2bf(bf2f(h[0]) * gelu_erf_grad(bf2f(uu[0]))), f2bf(bf2f(h[1]) * gelu_erf_grad(bf2f(uu[1]))),
                                                   f2bf(bf2f(h[2]) * gelu_erf_grad(bf2f(uu[2]))), f2bf(bf2f(h[3]) * gelu_erf_grad(bf2f(uu[3])))});
                }
            }
        } else {
            constexpr int FSTRIDE = BN * 4 + 16;  // fp32 image of 64 rows (one wave row) at a time
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (hf) __syncthreads();
                if (wr == hf) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            f32x4 v = acc[i][j] + b4[j];
                            if (p.round_bf16) v = (f32x4){round_bf16(v[0]), round_bf16(v[1]), round_bf16(v[2]), round_bf16(v[3])};
                            *(f32x4*)(smem + (i * 16 + fr) * FSTRIDE + (wc * 64 + j * 16 + fq * 4) * 4) = v;
                        }
                }
                __syncthreads();
#pragma unroll 2
                for (int u = 0; u < 64 * (BN / 4) / 256; ++u) {
                    const int slot = u * 256 + tid;
                    const int row = slot >> 5, c = slot & 31;
                    const int m = m0 + hf * 64 + row, n = n0 + c * 4;
                    if (m >= p.M || n >= p.N) continue;
                    f32x4 v = *(const f32x4*)(smem + row * FSTRIDE + c * 16);
                    const int64_t orow = omap(m);
                    if (p.residual) v += *(const f32x4*)(p.residual + orow * p.ldr + n);
                    if (p.rowmod) v += *(const f32x4*)(p.rowmod + (int64_t)(m % p.rowmod_period) * p.N + n);
                    if (p.out_scale != 0.f) v *= p.out_scale;
                    *(f32x4*)((float*)p.out + orow * p.ldo + n) = v;
                    if (p.out2) st_stream((bf16x4*)((bf16_t*)p.out2 + orow * p.ldo2 + n), (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])});
                }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wc * 64 + j * 16 + fq * 4;
            if (n >= p.N) continue;
            nt_epilogue<EPI>(p, omap, m, n, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN (grouped)
// ------------------------------------------------------------------------------------------------
struct TNArgs {
    vtGemmTN p[VT_TN_MAX_GROUP];
    int tile_start[VT_TN_MAX_GROUP + 1];
    int n;
};

// [64 m-rows][128 cols] bf16 tile, 256-B rows, 16 chunks; physical chunk = logical ^ fT(row)
__device__ __forceinline__ int swz_tn(int row) { return ((row & 3) | (((row >> 3) & 1) << 2)) << 1; }

__device__ __forceinline__ void stage_tn(const bf16_t* __restrict__ g, int64_t ld, int m0, int c0, int ncols,
                                         char* lds, int tid, int wave) {
    const int maxchunk = (ncols >> 3) - 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int slot = i * 256 + tid;
        const int row = slot >> 4;
        const int lc = (slot & 15) ^ swz_tn(row);
        int gc = (c0 >> 3) + lc;
        gc = gc < maxchunk ? gc : maxchunk;  // columns past the matrix edge: any in-bounds data (never stored)
        glds16(g + (int64_t)(m0 + row) * ld + gc * 8, lds + (i * 256 + wave * 64) * 16);
    }
}

// fragment for the 16 columns starting at `col` (multiple of 16), k-step base kb (0/32): lane (g = l>>4,
// i = l&15) gets tile[kb + 8g + 0..7][col + i]
__device__ __forceinline__ bf16x8 frag_tn(const char* lds, int col, int kb, int lane) {
    const int g = lane >> 4, lam = lane & 15;
    const int q = lam >> 2, pp = lam & 3;
    const int lchunk = (col >> 3) + (pp >> 1);
    const int r0 = kb + 8 * g + q, r1 = r0 + 4;
    const bf16x4 lo = lds_read_tr16(lds + r0 * 256 + ((lchunk ^ swz_tn(r0)) << 4) + ((pp & 1) << 3));
    const bf16x4 hi = lds_read_tr16(lds + r1 * 256 + ((lchunk ^ swz_tn(r1)) << 4) + ((pp & 1) << 3));
    return cat4(lo, hi);
}

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TNArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

    const int nwg = a.tile_start[a.n];
    const int sid = xcd_remap(blockIdx.x, nwg);
    int g = 0;
    while (g + 1 < a.n && sid >= a.tile_start[g + 1]) ++g;
    const vtGemmTN& p = a.p[g];
    const int local = sid - a.tile_start[g];
    const int tiles_q = (p.q_lim + BN - 1) / BN;
    const int p0 = (local / tiles_q) * BM, q0 = (local % tiles_q) * BN;

    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.M / BK;
    // LDS: [buffer 0: A | B][buffer 1: A | B]

    stage_tn(A, p.lda, 0, p0, p.P, smem, tid, wave);
    stage_tn(B, p.ldb, 0, q0, p.Q, smem + TILE_B, tid, wave);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            stage_tn(A, p.lda, (t + 1) * BK, p0, p.P, smem + (cur ^ 1) * 2 * TILE_B, tid, wave);
            stage_tn(B, p.ldb, (t + 1) * BK, q0, p.Q, smem + (cur ^ 1) * 2 * TILE_B + TILE_B, tid, wave);
        }
        const char* la = smem + cur * 2 * TILE_B;
        const char* lb = la + TILE_B;
        bf16x8 af[2][4], bfv[2][4];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bfv[kk][j] = frag_tn(lb, wc * 64 + j * 16, kk * 32, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i) af[kk][i] = frag_tn(la, wr * 64 + i * 16, kk * 32, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }

    // acc[i][j][r] = C[p0 + wr*64 + i*16 + (lane&15)][q0 + wc*64 + j*16 + (lane>>4)*4 + r]
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int pr = p0 + wr * 64 + i * 16 + fr;
        if (pr >= p.p_lim) continue;
        const int64_t orow = p.row_perm ? (int64_t)p.row_perm[pr] : (int64_t)pr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qc = q0 + wc * 64 + j * 16 + fq * 4;
            if (qc >= p.q_lim) continue;
            float* o = p.out + orow * p.ldo + qc;
            if (qc + 3 < p.q_lim && ((p.ldo & 3) == 0)) {
                *(f32x4*)o = acc[i][j];
            } else {
                for (int r = 0; r < 4 && qc + r < p.q_lim; ++r) o[r] = acc[i][j][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ skinny NT (M <= 64)
// The KV-cache decode loop of the AR prior multiplies a handful of token rows (M = batch, 1-64) with every weight matrix:
// weight-streaming, latency-bound work on which a 128- or 192-row tile wastes the chip (N / 128 workgroups, each walking
// all of K serially: 13-24 us per launch at M = 16).  Here a 4-wave workgroup owns 16 weight rows (N / 16 workgroups), the
// waves deal the 32-wide k-steps round-robin (so together they read each weight row in adjacent 64-B pieces), every lane
// keeps 4 steps of loads in flight, and the four partial 16 x M accumulators are summed through LDS in a fixed order.
// A-operand = weight rows, B-operand = token rows: a lane ends up with 4 consecutive n of one token, as in the tile kernels,
// so the shared epilogue applies unchanged.
template <int EPI, int MT>
__global__ __launch_bounds__(256) void gemm_nt_skinny_kernel(const vtGemmNT p) {
    __shared__ float red[4][MT * 16 * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int fr = lane & 15, fq = lane >> 4;
    int wrow = n0 + fr;
    wrow = wrow < p.N ? wrow : p.N - 1;
    const bf16_t* wp = (const bf16_t*)p.B + (int64_t)wrow * p.ldb + 8 * fq;
    const bf16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int r = t * 16 + fr;
        r = r < p.M ? r : p.M - 1;
        xp[t] = (const bf16_t*)p.A + (int64_t)r * p.lda + 8 * fq;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = p.K / 32;
    for (int s0 = wave; s0 < steps; s0 += 16) {
        bf16x8 wf[4], xf[4][MT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int ks = s0 + 4 * u;
            const int kk = (ks < steps ? ks : s0) * 32;          // past the end: re-read a valid piece, not accumulated
            wf[u] = *(const bf16x8*)(wp + kk);
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[u][t] = *(const bf16x8*)(xp[t] + kk);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
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
    if (m < p.M && m < MT * 16) {
        const f32x4 v = (*(const f32x4*)(&red[0][m * 16 + 4 * g]) + *(const f32x4*)(&red[1][m * 16 + 4 * g])) +
                        (*(const f32x4*)(&red[2][m * 16 + 4 * g]) + *(const f32x4*)(&red[3][m * 16 + 4 * g]));
        const int n = n0 + 4 * g;
        if (n < p.N) nt_epilogue<EPI>(p, RowMap{p.omap.grp, p.omap.stride, p.omap.off}, m, n, v);
    }
}

template <int EPI>
static void launch_skinny(const vtGemmNT& p, hipStream_t s) {
    const dim3 grid((p.N + 15) / 16), block(256);
    const int mt = (p.M + 15) / 16;
    switch (mt) {
        case 1: hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, 1>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, 2>), grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, 3>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((gemm_nt_skinny_kernel<EPI, 4>), grid, block, 0, s, p); break;
    }
}

}  // namespace

constexpr int VT_SPLITK_CTR_BYTES = 4096;   // arrival counters (one per 128x128 output tile) in front of the partial sums
extern "C" size_t vt_gemm_nt_splitk_workspace_bytes(void) { return VT_SPLITK_CTR_BYTES + (size_t)512 * BM * BN * 4; }   // automatic rule: tiles x split <= 2 x 256

int vt_gemm_nt192_launch(const vtGemmNT& p, hipStream_t s, int dbg, int half, int one_tile, int order);
int vt_gemm_tn192_launch(const vtGemmTN* ph, int n, hipStream_t s, int burst);
int vt_gemm192_init();
extern "C" int vt_gemm_nt(const vtGemmNT* ph, vtStream stream) {
    const vtGemmNT& p = *ph;
    VT_CHECK_ARG(p.A && p.B && p.out, "vt_gemm_nt: null operand");
    const int g_gemm_variant = p.tile;   // per call (vtGemmNT.tile); the library holds no tile setting of its own
    VT_CHECK_ARG(g_gemm_variant >= 0 && g_gemm_variant <= 31, "vt_gemm_nt: tile %d (0 auto, 1 = 128x128 2-deep ring, 16 = 128x128 4-deep ring, 2 = 192x192, 5 = 192x96, 6 = 192x192 one tile per workgroup, 7 = skinny M <= 64; 3/4/17/18 timing ablations; 19..31 tile order of the 192x192 kernel)", g_gemm_variant);
    VT_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0 && p.K % BK == 0, "vt_gemm_nt: K=%d must be a positive multiple of 64 (M=%d N=%d)", p.K, p.M, p.N);
    VT_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.lda >= p.K && p.ldb >= p.K, "vt_gemm_nt: lda/ldb must be >= K and multiples of 8");
    VT_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "vt_gemm_nt: A/B must be 16-byte aligned");
    VT_CHECK_ARG(p.ldo % 4 == 0, "vt_gemm_nt: ldo must be a multiple of 4");
    VT_CHECK_ARG(((uintptr_t)p.out & 15) == 0 && ((uintptr_t)p.bias & 15) == 0 && ((uintptr_t)p.out2 & 7) == 0 &&
                 ((uintptr_t)p.residual & 15) == 0 && ((uintptr_t)p.rowmod & 15) == 0 && ((uintptr_t)p.aux & 7) == 0,
                 "vt_gemm_nt: out/bias/residual/rowmod must be 16-byte aligned (out2/aux 8-byte)");
    VT_CHECK_ARG(!p.rowmod || p.N % 4 == 0, "vt_gemm_nt: rowmod needs N %% 4 == 0");
    if (p.epi == VT_EPI_BF16_GELU) VT_CHECK_ARG(p.out2 && p.ldo2 % 4 == 0, "vt_gemm_nt: GELU epilogue needs out2");
    if (p.epi == VT_EPI_BF16_DGELU) VT_CHECK_ARG(p.aux && p.ldaux % 4 == 0, "vt_gemm_nt: DGELU epilogue needs aux");
    VT_CHECK_ARG(p.out_scale == 0.f || p.epi == VT_EPI_F32, "vt_gemm_nt: out_scale only with VT_EPI_F32");
    VT_CHECK_ARG(!p.colsum_partial || (p.epi == VT_EPI_BF16_DGELU && p.N % 4 == 0),
                 "vt_gemm_nt: colsum_partial needs the DGELU epilogue and N %% 4 == 0 (it always runs on the 192-row tile kernel)");
    if (p.epi == VT_EPI_F32) {
        VT_CHECK_ARG(!p.residual || p.ldr % 4 == 0, "vt_gemm_nt: ldr must be a multiple of 4");
        VT_CHECK_ARG(!p.rowmod || p.rowmod_period > 0, "vt_gemm_nt: rowmod needs a period");
        VT_CHECK_ARG(!p.out2 || p.ldo2 % 4 == 0, "vt_gemm_nt: ldo2 must be a multiple of 4");
    } else {
        VT_CHECK_ARG(p.omap.grp == 0, "vt_gemm_nt: output row map only with VT_EPI_F32");
    }
    // auto dispatch (measured on MI355X: tools/gemm_bench.py, tools/gemm_disc_shapes.py): the 192x192 3-stage kernel is
    // ahead whenever its tiles fill whole rounds of the 256 CUs (every tokenizer shape: 256 / 768 / 1024 tiles).  Shapes
    // that leave its last round mostly empty (the discriminator's M = 8 x 1025, N = 384 / 1152: 86 or 258 tiles) finish
    // sooner on 128x128 tiles, two workgroups per CU.
    // M <= 64 (the AR prior's decode steps): the weight-streaming kernel, N / 16 workgroups
    if (g_gemm_variant == 7 || (g_gemm_variant == 0 && p.M <= 64 && !p.colsum_partial && p.N >= 64)) {
        VT_CHECK_ARG(p.M <= 64 && !p.colsum_partial, "vt_gemm_nt: tile 7 (skinny) needs M <= 64 and no colsum_partial (M=%d)", p.M);
        hipStream_t s = (hipStream_t)stream;
        switch (p.epi) {
            case VT_EPI_BF16: launch_skinny<VT_EPI_BF16>(p, s); break;
            case VT_EPI_BF16_GELU: launch_skinny<VT_EPI_BF16_GELU>(p, s); break;
            case VT_EPI_F32: launch_skinny<VT_EPI_F32>(p, s); break;
            case VT_EPI_BF16_DGELU: launch_skinny<VT_EPI_BF16_DGELU>(p, s); break;
            default: vt_set_error("vt_gemm_nt: unknown epilogue %d", p.epi); return VT_ERR_INVALID;
        }
        VT_CHECK_LAUNCH("vt_gemm_nt(skinny)");
        return VT_OK;
    }
    bool big = (g_gemm_variant >= 2 && g_gemm_variant != 7 && g_gemm_variant != 16) || p.colsum_partial;
    if (g_gemm_variant == 0 && !big && p.N >= 192 && p.M >= 192) {
        // cost in units of one full round of 192x192 tiles (256 workgroups, one per CU).  A partly filled last round of
        // that kernel costs a whole round; the 128x128 kernel runs two workgroups per CU (512 per round, a round ~1.05 of
        // a 192-round on equal work), and its last round is cheap while it leaves one workgroup per CU
        // (tools/gemm_disc_shapes.py, gemm_lastblock_shapes.py, gemm_fsq_shapes.py: the rule picks the faster kernel on
        // every shape measured there).
        const long t192 = (long)((p.M + 191) / 192) * ((p.N + 191) / 192), t128 = (long)((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
        const double c192 = (double)((t192 + 255) / 256);
        const long rem = t128 % 512;
        const double last = rem == 0 ? 0.0 : rem <= 256 ? 0.6 : 0.6 + 0.4 * (double)(rem - 256) / 256.0;
        const double c128 = 1.05 * ((double)(t128 / 512) + last);
        big = c192 <= c128;
    }
    if (big) {
        int rc = vt_gemm192_init();
        if (rc) return rc;
        // 19..31: tile order of the 192x192 kernel for A/B timing (19 = row-major list, 19 + W = column blocks of W tile columns); same results
        const int order = (g_gemm_variant >= 19 && g_gemm_variant <= 31) ? g_gemm_variant - 19 : -1;
        const int dbg = order >= 0 ? 0 : (g_gemm_variant == 3 || g_gemm_variant == 4) ? g_gemm_variant - 2 : g_gemm_variant >= 17 ? g_gemm_variant - 1 : g_gemm_variant >= 8 ? g_gemm_variant : 0;
        const int half = g_gemm_variant == 5;
        vt_gemm_nt192_launch(p, (hipStream_t)stream, dbg, half, g_gemm_variant == 6, order);
        VT_CHECK_LAUNCH("vt_gemm_nt(192)");
        return VT_OK;
    }
    NTArgs a;
    a.p = p;
    a.tiles_m = (p.M + BM - 1) / BM;
    a.tiles_n = (p.N + BN - 1) / BN;
    a.split = 1; a.part = nullptr; a.ctr = nullptr;
    const int tiles = a.tiles_m * a.tiles_n;
    a.col_block = 0;   // set below, once the number of workgroups per CU is known
    hipStream_t s = (hipStream_t)stream;
    // At most one workgroup per CU (one or two clips per GPU): the 4-deep ring hides the load latency that a co-resident
    // workgroup would otherwise cover (same MFMA order, bit-identical results; tile 16 forces it, tile 1 keeps the 2-deep ring)
    static const int n_cu = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    // Split K (only with a caller-supplied workspace): a launch that would leave most of the chip idle -- the N = 768 GEMMs of a
    // one- or two-clip step are 72 or 144 tiles with K up to 3072 -- gives each tile to `split` workgroups, each over a
    // contiguous share of the K-tiles; the last one to arrive adds the fp32 partial sums in split order and runs the epilogue.
    // Deterministic (run-to-run bit-identical), but NOT the unsplit kernel's summation order: results differ from tile 1 / 16 in
    // the last fp32 bits.  Automatic rule from tools/gemm_small_m.py on MI355X.
    const int nt_all = p.K / BK;
    if (p.splitk_ws && p.splitk != 1 && (g_gemm_variant == 0 || g_gemm_variant == 1 || g_gemm_variant == 16)) {
        int want = p.splitk;
        if (want == 0 && g_gemm_variant == 0 && tiles < n_cu && nt_all >= 24) {
            // the hand-off costs ~6 us (write-through stores, the counter, one read past the L2): K >= 1536 pays for it, K = 768 does not;
            // three partials are read back in one go, more would queue behind each other
            want = tiles * 3 <= 2 * n_cu ? 3 : 2;
        }
        if (want > nt_all) want = nt_all;
        if (want >= 2) {
            const size_t need = VT_SPLITK_CTR_BYTES + (size_t)tiles * want * BM * BN * 4;
            VT_CHECK_ARG(tiles <= VT_SPLITK_CTR_BYTES / 4 && (size_t)p.splitk_ws_bytes >= need && ((uintptr_t)p.splitk_ws & 255) == 0,
                         "vt_gemm_nt: split-K workspace too small or misaligned (%lld bytes, this launch needs %zu: %d tiles x %d; "
                         "vt_gemm_nt_splitk_workspace_bytes() covers every automatic choice)", (long long)p.splitk_ws_bytes, need, tiles, want);
            a.split = want;
            a.ctr = (unsigned*)p.splitk_ws;
            a.part = (float*)((char*)p.splitk_ws + VT_SPLITK_CTR_BYTES);
        }
    }
    const dim3 grid(tiles * a.split), block(256);
    const bool deep = g_gemm_variant == 16 || (g_gemm_variant == 0 && (int)grid.x <= n_cu && nt_all / a.split >= 4);
    {   // tile order: the tiles one XCD has in flight form a rectangle (vt_common.h); VT_GEMM_TILE_ORDER forces a width for whole-step A/B timing
        static const int order_env = [] { const char* e = getenv("VT_GEMM_TILE_ORDER"); return (e && *e) ? atoi(e) : -1; }();
        const int per_xcd = (deep ? 1 : 2) * n_cu / 8, chunk = ((int)grid.x + 7) / 8;
        a.col_block = order_env >= 0 ? (order_env < a.tiles_n ? order_env : 0) : vt_auto_col_block(a.tiles_n, per_xcd < chunk ? per_xcd : chunk);
    }
    if (deep) {
        static bool attr_set = false;
        constexpr int LDS4 = 4 * 2 * TILE_B;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_kernel<VT_EPI_BF16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<VT_EPI_BF16_GELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<VT_EPI_F32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_kernel<VT_EPI_BF16_DGELU, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS4);
            if (e != hipSuccess) {
                vt_set_error("vt_gemm_nt: hipFuncSetAttribute(%d bytes of LDS) failed: %s", LDS4, hipGetErrorString(e));
                return VT_ERR_LAUNCH;
            }
            attr_set = true;
        }
        switch (p.epi) {
            case VT_EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16, 4>), grid, block, LDS4, s, a); break;
            case VT_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16_GELU, 4>), grid, block, LDS4, s, a); break;
            case VT_EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_F32, 4>), grid, block, LDS4, s, a); break;
            case VT_EPI_BF16_DGELU: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16_DGELU, 4>), grid, block, LDS4, s, a); break;
            default: vt_set_error("vt_gemm_nt: unknown epilogue %d", p.epi); return VT_ERR_INVALID;
        }
        VT_CHECK_LAUNCH("vt_gemm_nt(128, 4-deep)");
        return VT_OK;
    }
    const size_t lds = 4 * TILE_B;
    switch (p.epi) {
        case VT_EPI_BF16: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16, 2>), grid, block, lds, s, a); break;
        case VT_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16_GELU, 2>), grid, block, lds, s, a); break;
        case VT_EPI_F32: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_F32, 2>), grid, block, lds, s, a); break;
        case VT_EPI_BF16_DGELU: hipLaunchKernelGGL((gemm_nt_kernel<VT_EPI_BF16_DGELU, 2>), grid, block, lds, s, a); break;
        default: vt_set_error("vt_gemm_nt: unknown epilogue %d", p.epi); return VT_ERR_INVALID;
    }
    VT_CHECK_LAUNCH("vt_gemm_nt");
    return VT_OK;
}

extern "C" int vt_gemm_tn_grouped(const vtGemmTN* ph, int32_t n, vtStream stream) {
    VT_CHECK_ARG(ph && n > 0 && n <= VT_TN_MAX_GROUP, "vt_gemm_tn_grouped: 1..%d problems", VT_TN_MAX_GROUP);
    TNArgs a;
    a.n = n;
    a.tile_start[0] = 0;
    const int g_gemm_variant = ph[0].tile;
    VT_CHECK_ARG(g_gemm_variant >= 0 && g_gemm_variant <= 7, "vt_gemm_tn_grouped: tile %d (0 auto, 1 = 128x128, 2 = 192x192, 7 = 192x192 without the register pipeline)", g_gemm_variant);
    bool big = g_gemm_variant != 1 && (g_gemm_variant < 3 || g_gemm_variant >= 5);  // auto: 192x192 tiles when every problem of the group is at least one tile
    for (int g = 0; g < n; ++g) {
        const vtGemmTN& p = ph[g];
        VT_CHECK_ARG(p.A && p.B && p.out, "vt_gemm_tn_grouped[%d]: null operand", g);
        VT_CHECK_ARG(p.M > 0 && p.M % BK == 0, "vt_gemm_tn_grouped[%d]: M=%d must be a multiple of 64 (pad rows with zeros)", g, p.M);
        VT_CHECK_ARG(p.P >= 8 && p.Q >= 8 && p.P % 8 == 0 && p.Q % 8 == 0, "vt_gemm_tn_grouped[%d]: P=%d Q=%d must be multiples of 8", g, p.P, p.Q);
        VT_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && p.lda >= p.P && p.ldb >= p.Q, "vt_gemm_tn_grouped[%d]: bad lda/ldb", g);
        VT_CHECK_ARG(((uintptr_t)p.A & 15) == 0 && ((uintptr_t)p.B & 15) == 0, "vt_gemm_tn_grouped[%d]: A/B must be 16-byte aligned", g);
        VT_CHECK_ARG(p.p_lim > 0 && p.p_lim <= p.P && p.q_lim > 0 && p.q_lim <= p.Q, "vt_gemm_tn_grouped[%d]: bad limits", g);
        a.p[g] = p;
        const int tp = (p.p_lim + BM - 1) / BM, tq = (p.q_lim + BN - 1) / BN;
        a.tile_start[g + 1] = a.tile_start[g] + tp * tq;
        if (g_gemm_variant == 0 && (p.p_lim < 192 || p.q_lim < 192)) big = false;
    }
    if (big) {
        int rc = vt_gemm192_init();
        if (rc) return rc;
        vt_gemm_tn192_launch(ph, n, (hipStream_t)stream, g_gemm_variant == 7);
        VT_CHECK_LAUNCH("vt_gemm_tn_grouped(192)");
        return VT_OK;
    }
    const dim3 grid(a.tile_start[n]), block(256);
    hipLaunchKernelGGL(gemm_tn_kernel, grid, block, 4 * TILE_B, (hipStream_t)stream, a);
    VT_CHECK_LAUNCH("vt_gemm_tn_grouped");
    return VT_OK;
}
