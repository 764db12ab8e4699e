// Second-generation bf16 MFMA GEMMs for gfx950: 192x192x64 tiles, one 512-thread workgroup per CU.
//
// Why 192: the tokenizer's GEMM extents are M = B*1536 = B*8*192 and N in {768, 1536, 2304, 3072} =
// {4, 8, 12, 16} x 192.  At 8 clips/GPU the N = 768 GEMMs (5 of the 8 per transformer block) are exactly
// 64 x 4 = 256 tiles = one per CU with no tail wave, where 128x128 tiles give 576 tiles on 512 slots.
//
// One workgroup per CU means nothing else hides the HBM/L2 latency, so the staging is a 3-deep ring of
// LDS buffers (3 x 48 KiB = 144 of the CU's 160 KiB) filled by 16-B LDS-DMA (global_load_lds) and ordered
// by a COUNTED s_waitcnt vmcnt(6) + a raw s_barrier per K-tile: the loads of K-tile t+2 are issued right
// after the barrier of tile t and stay in flight across the next barrier (a __syncthreads() would drain
// them).  8 waves = 2 per SIMD as 2(M) x 4(N), 96x48 outputs per wave = 6x3 accumulators of
// v_mfma_f32_16x16x32_bf16; the two waves of a SIMD interleave LDS reads with MFMAs.
//   NT: fragments by ds_read_b128 from [192][64] images (128-B rows, chunk ^= (row>>1)&7).
//   TN: fragments by ds_read_b64_tr_b16 from [64][192] images (384-B rows, low 3 chunk bits ^= f(row)).
#include "vt_common.h"
#include "vt_gemm_epilogue.h"

namespace {

constexpr int TM = 192, TN_ = 192, TK = 64;
constexpr int OP_BYTES = TM * TK * 2;      // 24 KiB per operand tile (both layouts)
constexpr int STAGE_BYTES = 2 * OP_BYTES;  // A | B
constexpr int NSTAGE = 3;

__device__ __forceinline__ void wait_vmcnt6() { asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
__device__ __forceinline__ void wait_vmcnt0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void raw_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ------------------------------------------------------------------------------------------------ NT
struct NT192Args {
    vtGemmNT p;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void stage_nt192(const bf16_t* __restrict__ g, int64_t ld, int row0, int nrows, int k0, unsigned lds, int tid, int wave) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int slot = i * 512 + tid;
        const int row = slot >> 3;
        const int lc = (slot & 7) ^ ((row >> 1) & 7);
        int gr = row0 + row;
        gr = gr < nrows ? gr : nrows - 1;
        glds16_asm(g + (int64_t)gr * ld + k0 + lc * 8, lds + (i * 512 + wave * 64) * 16);
    }
}

__device__ __forceinline__ bf16x8 frag_nt192(const char* lds, int row, int lchunk) {
    return *(const bf16x8*)(lds + row * 128 + ((lchunk ^ ((row >> 1) & 7)) << 4));
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt192_kernel(const NT192Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const vtGemmNT& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = a.tiles_m * a.tiles_n;
    const int sid = xcd_remap(blockIdx.x, nwg);
    const int m0 = (sid / a.tiles_n) * TM, n0 = (sid % a.tiles_n) * TN_;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / TK;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    stage_nt192(A, p.lda, m0, p.M, 0, sbase, tid, wave);
    stage_nt192(B, p.ldb, n0, p.N, 0, sbase + OP_BYTES, tid, wave);
    if (nt > 1) {
        stage_nt192(A, p.lda, m0, p.M, TK, sbase + STAGE_BYTES, tid, wave);
        stage_nt192(B, p.ldb, n0, p.N, TK, sbase + STAGE_BYTES + OP_BYTES, tid, wave);
    }
    const int fr = lane & 15, fq = lane >> 4;
    // byte offset of this lane's first A / B fragment inside an operand tile, for k-step 0 and 1.  The swizzle term
    // ((row>>1)&7) does not depend on the 16-row fragment index (16 rows = 8 swizzle periods), so fragment i sits at
    // +i*2048 bytes: an instruction immediate.
    const int arow = wm * 96 + fr, brow = wn * 48 + fr;
    const unsigned a_off0 = arow * 128 + ((fq ^ ((arow >> 1) & 7)) << 4), a_off1 = arow * 128 + (((4 + fq) ^ ((arow >> 1) & 7)) << 4);
    const unsigned b_off0 = brow * 128 + ((fq ^ ((brow >> 1) & 7)) << 4), b_off1 = brow * 128 + (((4 + fq) ^ ((brow >> 1) & 7)) << 4);
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) wait_vmcnt6(); else wait_vmcnt0();  // this thread's pieces of tile t have landed
        raw_barrier();                                        // everyone's have; everyone is done reading tile t-1
        if (t + 2 < nt) {
            int nx = cur + 2; nx = nx >= NSTAGE ? nx - NSTAGE : nx;
            stage_nt192(A, p.lda, m0, p.M, (t + 2) * TK, sbase + nx * STAGE_BYTES, tid, wave);
            stage_nt192(B, p.ldb, n0, p.N, (t + 2) * TK, sbase + nx * STAGE_BYTES + OP_BYTES, tid, wave);
        }
        // Fragment reads are hand-issued (inline asm) so that the LDS waits can be COUNTED: hipcc emits lgkmcnt(0)
        // for ds_read_b128 fragments in this loop.  LDS returns in order, so after issuing reads r0..r13 a
        // wait lgkmcnt(13 - k) means r0..rk have landed.  Each wait is followed by sched_barrier(0) so no MFMA is
        // hoisted above it.  Order: b0[0..2] a0[0..5] | b1[0..2] a1[0..1]   (14 in flight: 4-bit counter)
        const unsigned ta = sbase + cur * STAGE_BYTES;
        const unsigned a_k0 = ta + a_off0, a_k1 = ta + a_off1, b_k0 = ta + OP_BYTES + b_off0, b_k1 = ta + OP_BYTES + b_off1;
        bf16x8 a0[6], b0[3], a1[6], b1[3];
#define VT_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define VT_LGKM(n)                                            \
    __builtin_amdgcn_sched_barrier(0);                        \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0)
#define VT_ROW(accrow, bb, aa)                                                                   \
    accrow[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[0], aa, accrow[0], 0, 0, 0);          \
    accrow[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[1], aa, accrow[1], 0, 0, 0);          \
    accrow[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[2], aa, accrow[2], 0, 0, 0)
        VT_DSR(b0[0], b_k0, 0); VT_DSR(b0[1], b_k0, 2048); VT_DSR(b0[2], b_k0, 4096);
        VT_DSR(a0[0], a_k0, 0); VT_DSR(a0[1], a_k0, 2048); VT_DSR(a0[2], a_k0, 4096);
        VT_DSR(a0[3], a_k0, 6144); VT_DSR(a0[4], a_k0, 8192); VT_DSR(a0[5], a_k0, 10240);
        VT_DSR(b1[0], b_k1, 0); VT_DSR(b1[1], b_k1, 2048); VT_DSR(b1[2], b_k1, 4096);
        VT_DSR(a1[0], a_k1, 0); VT_DSR(a1[1], a_k1, 2048);
        VT_LGKM(10); VT_ROW(acc[0], b0, a0[0]);
        VT_LGKM(9);  VT_ROW(acc[1], b0, a0[1]);
        VT_LGKM(8);  VT_ROW(acc[2], b0, a0[2]);
        VT_LGKM(7);  VT_ROW(acc[3], b0, a0[3]);
        __builtin_amdgcn_sched_barrier(0);
        VT_DSR(a1[2], a_k1, 4096); VT_DSR(a1[3], a_k1, 6144);           // 6 + 2 in flight
        VT_LGKM(7);  VT_ROW(acc[4], b0, a0[4]);
        VT_LGKM(6);  VT_ROW(acc[5], b0, a0[5]);
        __builtin_amdgcn_sched_barrier(0);
        VT_DSR(a1[4], a_k1, 8192); VT_DSR(a1[5], a_k1, 10240);          // b1 x3, a1[0..3], + 2 = 9 in flight
        VT_LGKM(5);  VT_ROW(acc[0], b1, a1[0]);
        VT_LGKM(4);  VT_ROW(acc[1], b1, a1[1]);
        VT_LGKM(3);  VT_ROW(acc[2], b1, a1[2]);
        VT_LGKM(2);  VT_ROW(acc[3], b1, a1[3]);
        VT_LGKM(1);  VT_ROW(acc[4], b1, a1[4]);
        VT_LGKM(0);  VT_ROW(acc[5], b1, a1[5]);
        __builtin_amdgcn_sched_barrier(0);
#undef VT_DSR
#undef VT_LGKM
#undef VT_ROW
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }

    const RowMap omap{p.omap.grp, p.omap.stride, p.omap.off};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int m = m0 + wm * 96 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + wn * 48 + j * 16 + fq * 4;
            if (n >= p.N) continue;
            nt_epilogue<EPI>(p, omap, m, n, acc[i][j]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ TN
struct TN192Args {
    vtGemmTN p[VT_TN_MAX_GROUP];
    int tile_start[VT_TN_MAX_GROUP + 1];
    int n;
};

// [64 m-rows][192 cols] bf16 image, 384-B rows = 24 chunks; physical chunk = (lc & ~7) | ((lc & 7) ^ f(row))
__device__ __forceinline__ int swz_tn192(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

__device__ __forceinline__ void stage_tn192(const bf16_t* __restrict__ g, int64_t ld, int m0, int c0, int ncols, unsigned lds, int tid, int wave) {
    const int maxchunk = (ncols >> 3) - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int slot = i * 512 + tid;
        const int row = slot / 24;
        const int pc = slot - row * 24;
        const int lc = (pc & ~7) | ((pc & 7) ^ swz_tn192(row));
        int gc = (c0 >> 3) + lc;
        gc = gc < maxchunk ? gc : maxchunk;
        glds16_asm(g + (int64_t)(m0 + row) * ld + gc * 8, lds + (i * 512 + wave * 64) * 16);
    }
}

__device__ __forceinline__ bf16x8 frag_tn192(const char* lds, int col, int kb, int lane) {
    const int g = lane >> 4, lam = lane & 15;
    const int q = lam >> 2, pp = lam & 3;
    const int lc = (col >> 3) + (pp >> 1);
    const int r0 = kb + 8 * g + q, r1 = r0 + 4;
    const int c0 = (lc & ~7) | ((lc & 7) ^ swz_tn192(r0));
    const int c1 = (lc & ~7) | ((lc & 7) ^ swz_tn192(r1));
    const bf16x4 lo = lds_read_tr16(lds + r0 * 384 + (c0 << 4) + ((pp & 1) << 3));
    const bf16x4 hi = lds_read_tr16(lds + r1 * 384 + (c1 << 4) + ((pp & 1) << 3));
    return cat4(lo, hi);
}

__global__ __launch_bounds__(512, 2) void gemm_tn192_kernel(const TN192Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = a.tile_start[a.n];
    const int sid = xcd_remap(blockIdx.x, nwg);
    int g = 0;
    while (g + 1 < a.n && sid >= a.tile_start[g + 1]) ++g;
    const vtGemmTN& p = a.p[g];
    const int local = sid - a.tile_start[g];
    const int tiles_q = (p.q_lim + TN_ - 1) / TN_;
    const int p0 = (local / tiles_q) * TM, q0 = (local % tiles_q) * TN_;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.M / TK;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    stage_tn192(A, p.lda, 0, p0, p.P, sbase, tid, wave);
    stage_tn192(B, p.ldb, 0, q0, p.Q, sbase + OP_BYTES, tid, wave);
    if (nt > 1) {
        stage_tn192(A, p.lda, TK, p0, p.P, sbase + STAGE_BYTES, tid, wave);
        stage_tn192(B, p.ldb, TK, q0, p.Q, sbase + STAGE_BYTES + OP_BYTES, tid, wave);
    }
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) wait_vmcnt6(); else wait_vmcnt0();
        raw_barrier();
        if (t + 2 < nt) {
            int nx = cur + 2; nx = nx >= NSTAGE ? nx - NSTAGE : nx;
            stage_tn192(A, p.lda, (t + 2) * TK, p0, p.P, sbase + nx * STAGE_BYTES, tid, wave);
            stage_tn192(B, p.ldb, (t + 2) * TK, q0, p.Q, sbase + nx * STAGE_BYTES + OP_BYTES, tid, wave);
        }
        const char* la = smem + cur * STAGE_BYTES;
        const char* lb = la + OP_BYTES;
        // 36 transposed 8-byte reads per K-tile; at most 14 in flight (4-bit LGKM counter) ahead of the MFMAs
        bf16x8 af[2][6], bfv[2][3];
#pragma unroll
        for (int j = 0; j < 3; ++j) bfv[0][j] = frag_tn192(lb, wn * 48 + j * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) af[0][i] = frag_tn192(la, wm * 96 + i * 16, 0, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 0) {
                af[0][4] = frag_tn192(la, wm * 96 + 4 * 16, 0, lane);
                af[0][5] = frag_tn192(la, wm * 96 + 5 * 16, 0, lane);
            } else {
                bfv[1][i - 1] = frag_tn192(lb, wn * 48 + (i - 1) * 16, 32, lane);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[0][j], af[0][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 4; i < 6; ++i) {
            af[1][2 * (i - 4)] = frag_tn192(la, wm * 96 + (2 * (i - 4)) * 16, 32, lane);
            af[1][2 * (i - 4) + 1] = frag_tn192(la, wm * 96 + (2 * (i - 4) + 1) * 16, 32, lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[0][j], af[0][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < 2) af[1][4 + i] = frag_tn192(la, wm * 96 + (4 + i) * 16, 32, lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[1][j], af[1][i], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    }

    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pr = p0 + wm * 96 + i * 16 + fr;
        if (pr >= p.p_lim) continue;
        const int64_t orow = p.row_perm ? (int64_t)p.row_perm[pr] : (int64_t)pr;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int qc = q0 + wn * 48 + j * 16 + fq * 4;
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

}  // namespace

// Called by vt_gemm_nt / vt_gemm_tn_grouped (vt_gemm.hip) after argument validation.
int vt_gemm_nt192_launch(const vtGemmNT& p, hipStream_t s) {
    NT192Args a;
    a.p = p;
    a.tiles_m = (p.M + TM - 1) / TM;
    a.tiles_n = (p.N + TN_ - 1) / TN_;
    const dim3 grid(a.tiles_m * a.tiles_n), block(512);
    const size_t lds = NSTAGE * STAGE_BYTES;
    switch (p.epi) {
        case VT_EPI_BF16: hipLaunchKernelGGL(gemm_nt192_kernel<VT_EPI_BF16>, grid, block, lds, s, a); break;
        case VT_EPI_BF16_GELU: hipLaunchKernelGGL(gemm_nt192_kernel<VT_EPI_BF16_GELU>, grid, block, lds, s, a); break;
        case VT_EPI_F32: hipLaunchKernelGGL(gemm_nt192_kernel<VT_EPI_F32>, grid, block, lds, s, a); break;
        default: hipLaunchKernelGGL(gemm_nt192_kernel<VT_EPI_BF16_DGELU>, grid, block, lds, s, a); break;
    }
    return 0;
}

int vt_gemm_tn192_launch(const vtGemmTN* ph, int n, hipStream_t s) {
    TN192Args a;
    a.n = n;
    a.tile_start[0] = 0;
    for (int g = 0; g < n; ++g) {
        a.p[g] = ph[g];
        const int tp = (ph[g].p_lim + TM - 1) / TM, tq = (ph[g].q_lim + TN_ - 1) / TN_;
        a.tile_start[g + 1] = a.tile_start[g] + tp * tq;
    }
    hipLaunchKernelGGL(gemm_tn192_kernel, dim3(a.tile_start[n]), dim3(512), NSTAGE * STAGE_BYTES, s, a);
    return 0;
}

int vt_gemm192_init() {
    // 144 KiB of dynamic LDS exceeds the 64 KiB default: opt in once per kernel
    static bool done = false;
    if (done) return 0;
    const int lds = NSTAGE * STAGE_BYTES;
    hipError_t e = hipSuccess;
    e = hipFuncSetAttribute((const void*)gemm_nt192_kernel<VT_EPI_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192_kernel<VT_EPI_BF16_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192_kernel<VT_EPI_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192_kernel<VT_EPI_BF16_DGELU>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        vt_set_error("vt_gemm192_init: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return VT_ERR_LAUNCH;
    }
    done = true;
    return 0;
}
