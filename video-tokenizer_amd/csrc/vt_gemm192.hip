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
#include <stdlib.h>

#include "vt_common.h"
#include "vt_gemm_epilogue.h"

// -DVT_GEMM_STORE8 (A/B builds only, tools/ab_variant.sh): keep the 8-byte epilogue stores of rounds 1-3
#ifdef VT_GEMM_STORE8
constexpr bool kStore16 = false;
#else
constexpr bool kStore16 = true;
#endif

// Diagnostic build only (-DVT_GEMM_STAMPS, tools/gemm_stamps.sh): s_memtime stamps around the segments of one OUTPUT tile of the NT kernel,
// summed per wave: 0 main loop | 1 barrier + next tile's K-tile 0 issued | 2 accumulators -> LDS image | 3 barrier | 4 read-back + global stores |
// 5 barrier, K-tiles 1-2 issued, wait for K-tile 0, barrier | 6 fragments of K-tile 0 | 7 output tiles.  Read the shares, not the run time.
#ifdef VT_GEMM_STAMPS
__device__ unsigned long long g_nt_stamps[256][8][8];
#define VT_GSTAMP_DECL unsigned long long gs_t = __builtin_amdgcn_s_memtime(), gs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define VT_GSTAMP(i)                                                     \
    {                                                                    \
        __builtin_amdgcn_sched_barrier(0);                               \
        const unsigned long long gs_n = __builtin_amdgcn_s_memtime();    \
        gs_acc[i] += gs_n - gs_t;                                        \
        gs_t = gs_n;                                                     \
        __builtin_amdgcn_sched_barrier(0);                               \
    }
#define VT_GSTAMP_FLUSH                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 256)                     \
        for (int i_ = 0; i_ < 8; ++i_) g_nt_stamps[blockIdx.x][threadIdx.x >> 6][i_] = gs_acc[i_]
#else
#define VT_GSTAMP_DECL
#define VT_GSTAMP(i)
#define VT_GSTAMP_FLUSH
#endif

// -DVT_GEMM_RESIDUAL_IN_LOOP (A/B builds only): the fp32 epilogue of rounds 1-3, one residual load per trip of its store loop
#ifdef VT_GEMM_RESIDUAL_IN_LOOP
constexpr bool kResidualFirst = false;
#else
constexpr bool kResidualFirst = true;
#endif

namespace {

// A/B diagnostic (tools/r05_store_kind_ab.sh): which bf16 outputs of the epilogues leave by ORDINARY instead of streaming (nt) stores.
// bit 0 = the plain bf16 output (qkv forward, input gradients), bit 1 = the GELU epilogue's u, bit 2 = its g, bit 3 = the gelu' output.
#ifndef VT_GEMM_PLAIN_STORES
#define VT_GEMM_PLAIN_STORES 0
#endif
template <int BIT, typename T>
__device__ __forceinline__ void st_out(T* p, const T& v) {
    if constexpr ((VT_GEMM_PLAIN_STORES >> BIT) & 1) *p = v;
    else st_stream_any(p, v);
}

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
    int col_block;  // tile order: 0 = row-major list, W > 0 = column blocks of W tile columns, row-major inside a block (launch_nt192)
    int dbg;      // timing experiments only (vtGemmNT.tile 3/4/17/18): 1 = no LDS-DMA after the prologue, 2 = no MFMA/LDS reads, 16 = no bf16 output stores, 17 = output stores onto a cache-resident region
};

// Geometry of the two NT instantiations.  WN = waves along N (each wave owns 96 x 48 outputs):
//   WN = 4: 512 threads, 192 x 192 tile, 3-stage ring (144 KiB)  -> one workgroup per CU
//   WN = 2: 256 threads, 192 x  96 tile, 2-stage ring ( 72 KiB)  -> two workgroups per CU.  Measured (tools/
//           gemm_half_bench.py): within 0-8 % BEHIND the 192x192 kernel on every training shape.  Co-resident
//           workgroups start together and stay in phase (the dispatcher puts blocks b and b+256 on one CU,
//           tools/probes/placement_probe.hip), so their epilogues do not fall into each other's main loops, and a
//           forced start offset cost more than it recovered.  Kept as a tile option (vtGemmNT.tile = 5) and as
//           the record of that experiment; auto dispatch never picks it.
template <int WN>
struct NTGeo {
    static constexpr int THREADS = 128 * WN;
    static constexpr int TNW = 48 * WN;               // tile extent along N
    static constexpr int OPA = TM * TK * 2;           // A image bytes
    static constexpr int OPB = TNW * TK * 2;          // B image bytes
    static constexpr int STAGE = OPA + OPB;
    static constexpr int NST = WN == 4 ? 3 : 2;
    static constexpr int PA = TM * 8 / THREADS;       // 16-B DMA pieces per thread per K-tile, A rows
    static constexpr int PB = TNW * 8 / THREADS;      // ... B rows
    static constexpr int P = PA + PB;
};

// gelu(u) of the fc1 epilogue by table.  It is applied to a value that is ALREADY bf16 (the rounded pre-activation u), so it is
// a function of 16 bits: the 10 KiB of LDS above the 144-KiB ring hold bf16 gelu(u) for |u| in [2^-16, 16), both signs
// (2 x 20 x 128 entries), filled once per workgroup at kernel start by gelu_erf itself -- every entry is bit-for-bit what the
// arithmetic returns.  A wave whose 4-column group holds a value outside the range (~1e-5 of the elements) takes the arithmetic
// for that group (wave-uniform branch), so the output is bit-identical to the arithmetic everywhere (the 128x128 kernel keeps the
// arithmetic; tests compare the two with torch.equal).  Measured gain: 84.9 -> 82.2 us on fc1 forward at the step's shape
// (tools/gemm_gelu_epilogue.py) -- small, because the GELU epilogue's extra 17 us over the plain one is mostly its second 75-MB
// output, written by all 256 CUs in phase, not its ~17 VALU + v_rcp + v_exp per element.  The same table for gelu' (fp32, 16 KiB)
// was built for the fc2-dgrad epilogue and dropped: that instantiation has no register to spare (256 allocated) and spilled
// around the main loop.
template <int EPI>
struct GeluTab {
    static constexpr bool ON = EPI == VT_EPI_BF16_GELU;
    static constexpr int LO_EXP = 127 - 16;
    static constexpr int NEXP = 20;
    static constexpr int PER_SIGN = NEXP * 128;
    static constexpr int ENTRY = 2;
    static constexpr int BYTES = ON ? 2 * PER_SIGN * ENTRY : 0;
    // entry index of the bf16 bit pattern `b`, or >= 2 * PER_SIGN when out of range
    static __device__ __forceinline__ unsigned index(unsigned b) {
        const unsigned i = (b & 0x7FFFu) - (unsigned)(LO_EXP << 7);
        return i < (unsigned)PER_SIGN ? i + ((b >> 15) ? PER_SIGN : 0) : 0xFFFFu;
    }
};
__device__ __forceinline__ unsigned bf16_bits(bf16_t x) { return (unsigned)__builtin_bit_cast(unsigned short, x); }
__device__ __forceinline__ bf16_t bf16_from_bits(unsigned b) { return __builtin_bit_cast(bf16_t, (unsigned short)b); }

// The P 16-B-per-lane DMA pieces of a K-tile (pieces 0..PA-1 = A rows, PA..P-1 = B rows).  A lane's byte offset inside
// its operand's [tile rows][K] panel does not depend on the K-tile (row clamp and swizzle are per row), so it is computed
// once (nt192_piece_offsets) and every K-tile costs only the scalar advance of the two panel pointers: glds16_sv.
template <int WN>
__device__ __forceinline__ void nt192_piece_offsets(int64_t lda, int m0, int M, int64_t ldb, int n0, int N, int tid, unsigned (&off)[NTGeo<WN>::P]) {
    using G = NTGeo<WN>;
#pragma unroll
    for (int piece = 0; piece < G::P; ++piece) {
        const bool isA = piece < G::PA;
        const int i = isA ? piece : piece - G::PA;
        const int slot = i * G::THREADS + tid;
        const int row = slot >> 3;
        const int lc = (slot & 7) ^ ((row >> 1) & 7);
        const int r0 = isA ? m0 : n0, lim = isA ? M : N;
        int gr = r0 + row;
        gr = (gr < lim ? gr : lim - 1) - r0;          // rows past the matrix re-read its last row (never stored)
        off[piece] = (unsigned)((gr * (isA ? lda : ldb) + lc * 8) * 2);
    }
}
template <int WN>
__device__ __forceinline__ void nt192_stage_piece(const bf16_t* a_panel, const bf16_t* b_panel, const unsigned (&off)[NTGeo<WN>::P], unsigned lds,
                                                  int piece, int wave) {
    using G = NTGeo<WN>;
    const bool isA = piece < G::PA;
    const int i = isA ? piece : piece - G::PA;
    glds16_sv(isA ? a_panel : b_panel, off[piece], lds + (isA ? 0 : G::OPA) + (i * G::THREADS + wave * 64) * 16);
}

template <int EPI, int WN>
__global__ __launch_bounds__(128 * WN, 2) void gemm_nt192_kernel(const NT192Args a) {
    using G = NTGeo<WN>;
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const vtGemmNT& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = a.tiles_m * a.tiles_n;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    // Ring slot of K-tile t.  The 3-stage kernel is persistent (grid = min(tiles, CUs), workgroup b walks tiles b, b + grid,
    // ...): K-tile 0 of the NEXT output tile is DMA-ed into slot 2 before the epilogue of the current one -- the staged
    // epilogue image lives in slots 0-1 -- so its HBM latency runs under the epilogue's stores instead of in front of the
    // first MFMA (tools/gemm_epilogue_cost.py: ~3 us of the 6-20 us a tile pays outside its main loop).
    constexpr int ROT = G::NST == 3 ? 2 : 0;
    auto slot_of = [](int t) { return (t + ROT) % G::NST; };

    const int nt = p.K / TK;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int fr = lane & 15, fq = lane >> 4;
    // byte offset of this lane's first A / B fragment inside an operand tile, for k-step 0 and 1.  The swizzle term
    // ((row>>1)&7) does not depend on the 16-row fragment index (16 rows = 8 swizzle periods), so fragment i sits at
    // +i*2048 bytes: an instruction immediate.
    const int arow = wm * 96 + fr, brow = wn * 48 + fr;
    // k-step 1 is chunk 4 + fq: (4 + fq) ^ s = (fq ^ s) ^ 4, i.e. the k-step-0 ADDRESS with bit 6 flipped (ring slots and operand images are
    // multiples of 128 bytes) -- formed per K-tile from the k-step-0 address, so only two lane offsets live across the main loop.
    const unsigned a_off0 = arow * 128 + ((fq ^ ((arow >> 1) & 7)) << 4);
    const unsigned b_off0 = G::OPA + brow * 128 + ((fq ^ ((brow >> 1) & 7)) << 4);
    static_assert(G::OPA % 128 == 0 && G::STAGE % 128 == 0, "k-step 1 = k-step 0 ^ 64");

    // ---- software pipeline across K-tiles -------------------------------------------------------------------------
    // Tile t is multiplied out of a REGISTER set while the 18 fragment reads of tile t+1 are issued between its MFMA
    // rows into the other set, and the LDS-DMA of tile t+NST is in flight into the LDS buffer tile t just vacated.
    // (Without this, all 8 waves burst-read 144 KB of fragments after every barrier before any MFMA can issue: an
    // ablation with the DMA removed still ran at 88 % of the full kernel's time.)  Reads and DMAs are inline asm, so
    // every wait is explicit: vmcnt counts this thread's P DMA pieces per tile, lgkmcnt(0) closes a tile's reads
    // before the barrier that lets other waves overwrite that LDS buffer.
#define VT_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define VT_ROW(accrow, bb, aa)                                                                   \
    accrow[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[0], aa, accrow[0], 0, 0, 0);          \
    accrow[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[1], aa, accrow[1], 0, 0, 0);          \
    accrow[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[2], aa, accrow[2], 0, 0, 0);          \
    __builtin_amdgcn_sched_barrier(0)
    // compute from set C (a0,b0 = k-step 0; a1,b1 = k-step 1) while prefetching the tile at LDS address `nb` into set N
#define VT_DMA_ALL                                                                                                   \
    if (dma_tile >= 0) {                                                                                             \
        const bf16_t* ap_ = Apanel + dma_tile * TK;                                                                  \
        const bf16_t* bp_ = Bpanel + dma_tile * TK;                                                                  \
        _Pragma("unroll") for (int k_ = 0; k_ < G::P; ++k_) nt192_stage_piece<WN>(ap_, bp_, poff, dma_dst, k_, wave); \
    }
#define VT_STEP(Ca0, Cb0, Ca1, Cb1, Na0, Nb0, Na1, Nb1, nb, pf)                                                      \
    {                                                                                                                \
        const unsigned na0 = (nb) + a_off0, na1 = na0 ^ 64u, nb0 = (nb) + b_off0, nb1 = nb0 ^ 64u;                   \
        if (pf) { VT_DSR(Nb0[0], nb0, 0); VT_DSR(Nb0[1], nb0, 2048); }                                               \
        VT_ROW(acc[0], Cb0, Ca0[0]);                                                                                 \
        if (pf) { VT_DSR(Nb0[2], nb0, 4096); VT_DSR(Na0[0], na0, 0); }                                               \
        VT_ROW(acc[1], Cb0, Ca0[1]);                                                                                 \
        if (pf) { VT_DSR(Na0[1], na0, 2048); VT_DSR(Na0[2], na0, 4096); }                                            \
        VT_ROW(acc[2], Cb0, Ca0[2]);                                                                                 \
        if (pf) { VT_DSR(Na0[3], na0, 6144); VT_DSR(Na0[4], na0, 8192); }                                            \
        VT_ROW(acc[3], Cb0, Ca0[3]);                                                                                 \
        if (pf) { VT_DSR(Na0[5], na0, 10240); VT_DSR(Nb1[0], nb1, 0); }                                              \
        VT_ROW(acc[4], Cb0, Ca0[4]);                                                                                 \
        if (pf) { VT_DSR(Nb1[1], nb1, 2048); VT_DSR(Nb1[2], nb1, 4096); }                                            \
        VT_ROW(acc[5], Cb0, Ca0[5]);                                                                                 \
        VT_DMA_ALL;                                                                                                  \
        if (pf) { VT_DSR(Na1[0], na1, 0); }                                                                          \
        VT_ROW(acc[0], Cb1, Ca1[0]);                                                                                 \
        if (pf) { VT_DSR(Na1[1], na1, 2048); }                                                                       \
        VT_ROW(acc[1], Cb1, Ca1[1]);                                                                                 \
        if (pf) { VT_DSR(Na1[2], na1, 4096); }                                                                       \
        VT_ROW(acc[2], Cb1, Ca1[2]);                                                                                 \
        if (pf) { VT_DSR(Na1[3], na1, 6144); }                                                                       \
        VT_ROW(acc[3], Cb1, Ca1[3]);                                                                                 \
        if (pf) { VT_DSR(Na1[4], na1, 8192); }                                                                       \
        VT_ROW(acc[4], Cb1, Ca1[4]);                                                                                 \
        if (pf) { VT_DSR(Na1[5], na1, 10240); }                                                                      \
        VT_ROW(acc[5], Cb1, Ca1[5]);                                                                                 \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
    }
    unsigned poff[G::P];
    const bf16_t *Apanel, *Bpanel;   // wave-uniform: rows m0.. of A / n0.. of B, K-tile 0
    int m0, n0;
    bool poff_full = false;          // poff holds the offsets of a tile that lies fully inside the matrix (no row is clamped): the same for every such tile
    auto set_tile = [&](int it) {
        const int sid = xcd_remap(it, nwg);
        int tm, tn;   // an XCD's contiguous chunk of the list is a rectangle of tiles: fewer distinct operand panels among the tiles in flight on its L2
        vt_tile_of(sid, a.tiles_m, a.tiles_n, a.col_block, tm, tn);
        m0 = __builtin_amdgcn_readfirstlane(tm) * TM, n0 = __builtin_amdgcn_readfirstlane(tn) * G::TNW;
        const bool full = m0 + TM <= p.M && n0 + G::TNW <= p.N;
        if (!(full && poff_full)) {
            int tid_s = tid;
            asm volatile("" : "+v"(tid_s));  // opaque: the row / chunk terms of the offsets are recomputed when needed instead of living (spilled) across the main loop
            nt192_piece_offsets<WN>(p.lda, m0, p.M, p.ldb, n0, p.N, tid_s, poff);
            poff_full = full;
        }
        Apanel = A + (int64_t)m0 * p.lda;
        Bpanel = B + (int64_t)n0 * p.ldb;
    };
    auto issue_tile = [&](int tile) {  // LDS-DMA of K-tile `tile` into its ring slot
        const unsigned dst = sbase + slot_of(tile) * G::STAGE;
#pragma unroll
        for (int k = 0; k < G::P; ++k) nt192_stage_piece<WN>(Apanel + tile * TK, Bpanel + tile * TK, poff, dst, k, wave);
    };
    // make tile `nx` visible to every wave (its DMA landed everywhere) and recycle the buffer of tile nx-1, whose
    // fragments every wave already holds in registers, for tile nx-1+NST.  Its P DMA pieces are issued as one block in
    // the middle of the following tile body (VT_DMA_ALL), where the partner wave of the SIMD keeps the matrix pipe
    // busy.  With 3 stages one younger tile stays in flight across the barrier.
    auto sync_for = [&](int nx) {
        if (G::NST == 3 && nx + 1 < nt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::P) : "memory");
        else wait_vmcnt0();
        raw_barrier();
    };

    set_tile(blockIdx.x);
    issue_tile(0);
    using GT = GeluTab<EPI>;
    constexpr bool TABLE = GT::ON && WN == 4;
    [[maybe_unused]] char* const tab = smem + G::NST * G::STAGE;
    if constexpr (TABLE) {
        // visible to every wave long before the first epilogue: the main loop's barriers lie in between
        for (int e = tid; e < 2 * GT::PER_SIGN; e += G::THREADS) {
            const unsigned sgn = e >= GT::PER_SIGN ? 1u : 0u;
            const float x = bf2f(bf16_from_bits((sgn << 15) | (unsigned)(e - (int)sgn * GT::PER_SIGN + (GT::LO_EXP << 7))));
            ((bf16_t*)tab)[e] = f2bf(gelu_erf(x));
        }
    }
    if (a.dbg >= 8 && ((blockIdx.x >> 3) & 1))          // timing experiment (vtGemmNT.tile 8..15): every other CU of an XCD starts 1..8 us late
        for (int i = 0; i < a.dbg - 7; ++i) __builtin_amdgcn_s_sleep(32);
    constexpr bool PERSIST = WN == 4;   // the 192x96 experiment stays one tile per workgroup
    VT_GSTAMP_DECL;
    for (int it = blockIdx.x; it < nwg; it = PERSIST ? it + (int)gridDim.x : nwg) {
#ifdef VT_GEMM_STAMPS
    gs_acc[7] += 1;
    if (it == (int)blockIdx.x) gs_t = __builtin_amdgcn_s_memtime();
#endif
    // K-tile 0 of this output tile is already in flight (issued above, or before the previous tile's epilogue)
    // (zeroed by instructions with an inline constant: as plain C++ the compiler keeps a zero register PAIR alive across the whole
    // persistent loop to copy from, and spills it in the instantiations that sit at the register limit)
    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            typedef __attribute__((ext_vector_type(2))) float f32x2_;
            f32x2_ lo, hi;
            asm volatile("v_mov_b64 %0, 0" : "=v"(lo));
            asm volatile("v_mov_b64 %0, 0" : "=v"(hi));
            acc[i][j] = (f32x4){lo[0], lo[1], hi[0], hi[1]};
        }
    if (it != (int)blockIdx.x) __syncthreads();       // the previous epilogue's LDS image (slots 0-1) has been read back
    if (nt > 1) issue_tile(1);
    if (G::NST > 2 && nt > 2) issue_tile(2);
    if (G::NST > 2 && nt > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::P) : "memory");
    else if (nt > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::P) : "memory");
    else wait_vmcnt0();
    raw_barrier();
    VT_GSTAMP(5);
    bf16x8 xa0[6], xb0[3], xa1[6], xb1[3], ya0[6], yb0[3], ya1[6], yb1[3];
    {   // fragments of tile 0
        const unsigned s0 = sbase + slot_of(0) * G::STAGE;
        const unsigned na0 = s0 + a_off0, na1 = na0 ^ 64u, nb0 = s0 + b_off0, nb1 = nb0 ^ 64u;
        VT_DSR(xb0[0], nb0, 0); VT_DSR(xb0[1], nb0, 2048); VT_DSR(xb0[2], nb0, 4096);
        VT_DSR(xa0[0], na0, 0); VT_DSR(xa0[1], na0, 2048); VT_DSR(xa0[2], na0, 4096);
        VT_DSR(xa0[3], na0, 6144); VT_DSR(xa0[4], na0, 8192); VT_DSR(xa0[5], na0, 10240);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        VT_DSR(xb1[0], nb1, 0); VT_DSR(xb1[1], nb1, 2048); VT_DSR(xb1[2], nb1, 4096);
        VT_DSR(xa1[0], na1, 0); VT_DSR(xa1[1], na1, 2048); VT_DSR(xa1[2], na1, 4096);
        VT_DSR(xa1[3], na1, 6144); VT_DSR(xa1[4], na1, 8192); VT_DSR(xa1[5], na1, 10240);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    VT_GSTAMP(6);
    for (int t = 0;;) {
        // tile t from set X, prefetch tile t+1 into set Y.  The prefetch is unconditional (branch-free tile body, one
        // body per register set): after the last tile it re-reads a valid, quiescent LDS buffer into the unused set.
        {
            const bool more = t + 1 < nt;
            if (more) sync_for(t + 1);
            const unsigned nb = sbase + slot_of(more ? t + 1 : t) * G::STAGE;
            const int dma_tile = (t + G::NST < nt && a.dbg != 1) ? t + G::NST : -1;  // goes into the buffer tile t just vacated
            const unsigned dma_dst = sbase + slot_of(t) * G::STAGE;
            if (a.dbg != 2) VT_STEP(xa0, xb0, xa1, xb1, ya0, yb0, ya1, yb1, nb, true)
            if (++t == nt) break;
        }
        {   // tile t from set Y, prefetch tile t+1 into set X
            const bool more = t + 1 < nt;
            if (more) sync_for(t + 1);
            const unsigned nb = sbase + slot_of(more ? t + 1 : t) * G::STAGE;
            const int dma_tile = (t + G::NST < nt && a.dbg != 1) ? t + G::NST : -1;
            const unsigned dma_dst = sbase + slot_of(t) * G::STAGE;
            if (a.dbg != 2) VT_STEP(ya0, yb0, ya1, yb1, xa0, xb0, xa1, xb1, nb, true)
            if (++t == nt) break;
        }
    }
#undef VT_DSR
#undef VT_ROW
#undef VT_STEP
#undef VT_DMA_ALL

    // every wave is done with the ring: start the next output tile's K-tile 0 into slot 2 (the epilogue below only uses
    // slots 0-1), then write this tile out
    const int em0 = m0, en0 = n0;
    VT_GSTAMP(0);
    raw_barrier();
    if constexpr (PERSIST) {
        if (it + (int)gridDim.x < nwg) {
            set_tile(it + gridDim.x);
            if (a.dbg != 1) issue_tile(0);
        }
    }
    VT_GSTAMP(1);

    if constexpr (EPI != VT_EPI_F32) {
        if ((p.N & 3) == 0) {
            // the epilogue's per-thread coordinates are recomputed per output tile: hoisted out of the persistent loop they would
            // sit in registers across the main loop, which has none to spare (256 allocated, spill-free)
            int tid_e = tid;
            if constexpr (EPI != VT_EPI_BF16_DGELU) asm volatile("" : "+v"(tid_e));   // (the gelu' instantiation sits at 256 registers and spills WITH it)
            const int fr_e = tid_e & 15, fq_e = (tid_e & 63) >> 4;
            // bf16 outputs leave through LDS: after the swapped MFMA a lane owns 4 consecutive columns of 16 different
            // rows, so a direct store instruction touches 16 cache lines with 32 B each (the K -> 0 intercept of
            // tools/gemm_epilogue_cost.py: 2.4 TB/s of stores).  Staged as bf16(acc + bias) in a [192][TNW] image (row
            // stride +8 B: the 16 rows of a ds_write_b64 group fall on 16 different bank pairs) and read back row-major,
            // a store instruction writes 512 contiguous bytes.  GELU / gelu' are applied after the read-back, on the
            // bf16-rounded value (autocast order).
            constexpr int UPR = G::TNW / 4;             // 8-B units per tile row
            constexpr int STRIDE = G::TNW * 2 + 8;      // bytes
            constexpr int RL = G::THREADS / UPR;        // gelu' path: row lanes (10); RL * UPR of the threads are active
            constexpr int NIT = (TM + RL - 1) / RL;
            [[maybe_unused]] bf16x4 uu_pre[NIT];
            if constexpr (EPI == VT_EPI_BF16_DGELU) {
                // the pre-activations this thread will need after the read-back: all NIT loads go out now, so their HBM
                // latency runs under the staging writes and the barrier instead of 20 times in the store loop
                const int c = tid_e % UPR, rl = tid_e / UPR;
                const int n = en0 + c * 4;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int row = it * RL + rl;
                    const int m = em0 + row;
                    const bool ok = rl < RL && n < p.N && row < TM && m < p.M;
                    uu_pre[it] = ok ? ld_stream((const bf16x4*)((const bf16_t*)p.aux + (int64_t)m * p.ldaux + n)) : (bf16x4){f2bf(0.f), f2bf(0.f), f2bf(0.f), f2bf(0.f)};
                }
            }
            f32x4 b4[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = en0 + wn * 48 + j * 16 + fq_e * 4;
                b4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 v = acc[i][j] + b4[j];
                    *(bf16x4*)(smem + (wm * 96 + i * 16 + fr_e) * STRIDE + (wn * 12 + j * 4 + fq_e) * 8) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
                }
            VT_GSTAMP(2);
            __syncthreads();
            VT_GSTAMP(3);
            if constexpr (EPI == VT_EPI_BF16_DGELU) {
                // a thread keeps ONE 4-column group and walks the rows (48 lanes cover a 384-B row, the rest of the wave the
                // next row), so the column sums of the rounded output -- the bias gradient of the Linear whose
                // pre-activation is `aux` -- fall out of the epilogue instead of a separate pass over M x N
                const int c = tid_e % UPR, rl = tid_e / UPR;
                const int n = en0 + c * 4;
                f32x4 cs = {0.f, 0.f, 0.f, 0.f};
                if (rl < RL && n < p.N) {
#pragma unroll
                    for (int it = 0; it < NIT; ++it) {
                        const int row = it * RL + rl;
                        const int m = em0 + row;
                        if (row < TM && m < p.M) {
                            const bf16x4 h = *(const bf16x4*)(smem + row * STRIDE + c * 8);
                            const bf16x4 uu = uu_pre[it];
                            const bf16x4 r = {f2bf(bf2f(h[0]) * gelu_erf_grad(bf2f(uu[0]))), f2bf(bf2f(h[1]) * gelu_erf_grad(bf2f(uu[1]))),
                                              f2bf(bf2f(h[2]) * gelu_erf_grad(bf2f(uu[2]))), f2bf(bf2f(h[3]) * gelu_erf_grad(bf2f(uu[3])))};
                            st_out<3>((bf16x4*)((bf16_t*)p.out + (int64_t)m * p.ldo + n), r);
                            cs += (f32x4){bf2f(r[0]), bf2f(r[1]), bf2f(r[2]), bf2f(r[3])};
                        }
                    }
                }
                if (p.colsum_partial) {
                    float* red = (float*)(smem + TM * STRIDE);       // [RL][TNW], behind the staged image
                    if (rl < RL) *(f32x4*)(red + rl * G::TNW + c * 4) = cs;
                    __syncthreads();
                    if (tid_e < G::TNW && en0 + tid_e < p.N) {
                        float sum = 0.f;
#pragma unroll
                        for (int r = 0; r < RL; ++r) sum += red[r * G::TNW + tid_e];
                        p.colsum_partial[(int64_t)(em0 / TM) * p.N + en0 + tid_e] = sum;
                    }
                }
                continue;
            }
            // Round 4: 16 bytes per lane on the way out (two 8-byte image reads, ONE global_store_dwordx4: a wave instruction writes 1 KiB).
            // 8-byte accesses run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH, visibility table) and the epilogue is what a round
            // of this kernel pays outside its main loop (6-13 us, all 256 CUs storing at once).  Needs 8-column alignment of the output.
            if (kStore16 && a.dbg == 0 && (p.N & 7) == 0 && (p.ldo & 7) == 0 && (EPI != VT_EPI_BF16_GELU || (p.ldo2 & 7) == 0)) {
                constexpr int UPR16 = G::TNW / 8;                        // 16-B units per tile row
                constexpr int NIT16 = TM * UPR16 / G::THREADS, RB16 = 3;
                static_assert(NIT16 % RB16 == 0, "read-back batches");
                // WN = 4: a thread's 9 pieces are 3 row passes of 64 rows x 3 column groups of 64.  Lanes 8r..8r+7 of a wave instruction cover one
                // 128-byte line of row r (8 full lines per instruction), and every address is one per-thread base plus compile-time terms (the
                // slot / 24 mapping, kept for the 192x96 experiment, costs ~10 integer instructions per piece).
                auto piece_row = [&](int u) { if constexpr (WN == 4) return (tid_e >> 3) + (u / 3) * 64; else return (u * G::THREADS + tid_e) / UPR16; };
                auto piece_col = [&](int u) { if constexpr (WN == 4) return (tid_e & 7) * 8 + (u % 3) * 64; else return ((u * G::THREADS + tid_e) % UPR16) * 8; };
                auto piece_src = [&](int u) { return (const char*)smem + piece_row(u) * STRIDE + piece_col(u) * 2; };     // 8-byte aligned (row stride 392)
#pragma unroll 1
                for (int it0 = 0; it0 < NIT16; it0 += RB16) {
                    bf16x8 hh[RB16];
#pragma unroll
                    for (int u = 0; u < RB16; ++u) {
                        hh[u] = cat4(*(const bf16x4*)piece_src(it0 + u), *(const bf16x4*)(piece_src(it0 + u) + 8));
                    }
#pragma unroll
                    for (int u = 0; u < RB16; ++u) {
                        const int m = em0 + piece_row(it0 + u), n = en0 + piece_col(it0 + u);
                        if (m >= p.M || n >= p.N) continue;
                        const bf16x8 h = hh[u];
                        st_out<(EPI == VT_EPI_BF16_GELU ? 1 : 0)>((bf16x8*)((bf16_t*)p.out + (int64_t)m * p.ldo + n), h);
                        if constexpr (EPI == VT_EPI_BF16_GELU) {
                            bf16x8 gl;
                            bool looked_up = false;
                            if constexpr (TABLE) {
                                unsigned idx[8], any = 0;
#pragma unroll
                                for (int e = 0; e < 8; ++e) { idx[e] = GT::index(bf16_bits(h[e])); any |= idx[e]; }
                                if (__builtin_amdgcn_ballot_w64((any & 0x8000u) != 0) == 0) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) gl[e] = ((const bf16_t*)tab)[idx[e]];
                                    looked_up = true;
                                }
                            }
                            if (!looked_up) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) gl[e] = f2bf(gelu_erf(bf2f(h[e])));
                            }
                            st_out<2>((bf16x8*)((bf16_t*)p.out2 + (int64_t)m * p.ldo2 + n), gl);
                        }
                    }
                }
                VT_GSTAMP(4);
                if (it + (int)gridDim.x >= nwg) { VT_GSTAMP_FLUSH; }
                continue;
            }
            // read-back in batches of RB image reads followed by their stores: with one read in flight per store (the compiler's order for
            // the plain loop) every store waits a full LDS round trip
            // timing ablation (tile 18): every store of this workgroup lands on ONE tile-sized, cache-resident region (wave-uniform shift)
            const int64_t dbg_shift = a.dbg == 17 ? ((int64_t)((int)(blockIdx.x & 63) * TM - em0) * p.ldo - en0) : 0;
            constexpr int NITS = TM * UPR / G::THREADS, RB = 3;
            static_assert(NITS % RB == 0, "read-back batches");
#pragma unroll 1
            for (int it0 = 0; it0 < NITS; it0 += RB) {
                bf16x4 hh[RB];
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int slot = (it0 + u) * G::THREADS + tid_e;
                    const int row = slot / UPR, c = slot - row * UPR;
                    hh[u] = *(const bf16x4*)(smem + row * STRIDE + c * 8);
                }
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int slot = (it0 + u) * G::THREADS + tid_e;
                    const int row = slot / UPR, c = slot - row * UPR;
                    const int m = em0 + row, n = en0 + c * 4;
                    if (m >= p.M || n >= p.N) continue;
                    const bf16x4 h = hh[u];
                    if (a.dbg == 16 && bf16_bits(h[0]) != 0x7FC1u) continue;          // timing ablation (tile 17): no output stores
                    bf16_t* o = (bf16_t*)p.out + dbg_shift + (int64_t)m * p.ldo + n;
                    st_out<(EPI == VT_EPI_BF16_GELU ? 1 : 0)>((bf16x4*)o, h);
                    if constexpr (EPI == VT_EPI_BF16_GELU) {
                        bf16x4 gl;
                        bool looked_up = false;
                        if constexpr (TABLE) {
                            const unsigned i0 = GT::index(bf16_bits(h[0])), i1 = GT::index(bf16_bits(h[1])), i2 = GT::index(bf16_bits(h[2])), i3 = GT::index(bf16_bits(h[3]));
                            if (__builtin_amdgcn_ballot_w64(((i0 | i1 | i2 | i3) & 0x8000u) != 0) == 0) {
                                gl = (bf16x4){((const bf16_t*)tab)[i0], ((const bf16_t*)tab)[i1], ((const bf16_t*)tab)[i2], ((const bf16_t*)tab)[i3]};
                                looked_up = true;
                            }
                        }
                        if (!looked_up) gl = (bf16x4){f2bf(gelu_erf(bf2f(h[0]))), f2bf(gelu_erf(bf2f(h[1]))), f2bf(gelu_erf(bf2f(h[2]))), f2bf(gelu_erf(bf2f(h[3])))};
                        st_out<2>((bf16x4*)((bf16_t*)p.out2 + dbg_shift + (int64_t)m * p.ldo2 + n), gl);
                    }
                }
            }
            continue;
        }
    }
    const RowMap omap{p.omap.grp, p.omap.stride, p.omap.off};
    // (the lane's fragment coordinates again from an opaque copy of the thread id: the kernel-entry `fr` / `fq` would otherwise stay live -- spilled -- across the main loop)
    int tid_x = tid;
    asm volatile("" : "+v"(tid_x));
    const int fr_x = tid_x & 15, fq_x = (tid_x & 63) >> 4;
    if constexpr (EPI == VT_EPI_F32 && WN == 4) {
        if ((p.N & 3) == 0) {
            // fp32 outputs leave through LDS as well, 96 rows (one wave row) at a time: [96][192] fp32 image, row stride
            // +16 B.  Read back row-major, a wave's load / store instruction covers 768 contiguous bytes of the residual and
            // of the output instead of sixteen 64-byte pieces.  The fp32 output is the residual stream, which the next
            // kernel (a LayerNorm) reads straight away: ordinary stores measured better than streaming ones here.
            constexpr int FSTRIDE = TN_ * 4 + 16;
            f32x4 b4[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int n = en0 + wn * 48 + j * 16 + fq_x * 4;
                b4[j] = (p.bias && n < p.N) ? *(const f32x4*)(p.bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                if (hf) __syncthreads();
                // The residual rows of this half are requested FIRST, all NU of them: their latency runs under the staging writes and the
                // barrier.  (One load per trip of a rolled loop left 8 KiB in flight per CU -- 2 MB on the chip -- and every trip paid a full
                // memory round trip: the fp32 epilogue cost 11 us more than the bf16 one on the same product.)
                constexpr int NU = 96 * 48 / G::THREADS;
                // output row of tile row d without a division per piece: the tile starts at group q0, row r0 of the row map, and with
                // grp >= 192 it crosses at most one group boundary (other maps take the rolled loop below)
                const bool cheap_map = kResidualFirst && (omap.grp == 0 || omap.grp >= TM);
                const int q0 = omap.grp ? em0 / omap.grp : 0, r0 = omap.grp ? em0 - q0 * omap.grp : em0;
                auto orow_of = [&](int d) -> int64_t {
                    const int r = r0 + d;
                    if (omap.grp == 0) return r;
                    const bool wrap = r >= omap.grp;
                    return (int64_t)(q0 + (wrap ? 1 : 0)) * omap.stride + omap.off + (wrap ? r - omap.grp : r);
                };
                constexpr int RBF = 3;             // pieces per batch: one batch of loads is in flight while the previous one is added and stored
                static_assert(NU % RBF == 0, "residual batches");
                auto load_res = [&](int u) -> f32x4 {
                    const int slot = u * G::THREADS + tid_x;
                    const int row = slot / 48, c = slot - row * 48;
                    const int m = em0 + hf * 96 + row, n = en0 + c * 4;
                    return (p.residual && m < p.M && n < p.N) ? *(const f32x4*)(p.residual + orow_of(hf * 96 + row) * p.ldr + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
                };
                f32x4 res[RBF];
                if (cheap_map) {
#pragma unroll
                    for (int u = 0; u < RBF; ++u) res[u] = load_res(u);
                }
                if (wm == hf) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            f32x4 v = acc[i][j] + b4[j];
                            if (p.round_bf16) v = (f32x4){round_bf16(v[0]), round_bf16(v[1]), round_bf16(v[2]), round_bf16(v[3])};
                            *(f32x4*)(smem + (i * 16 + fr_x) * FSTRIDE + (wn * 48 + j * 16 + fq_x * 4) * 4) = v;
                        }
                }
                __syncthreads();
                if (cheap_map) {
#pragma unroll
                    for (int u0 = 0; u0 < NU; u0 += RBF) {
                        f32x4 nxt[RBF];
                        if (u0 + RBF < NU) {
#pragma unroll
                            for (int u = 0; u < RBF; ++u) nxt[u] = load_res(u0 + RBF + u);
                        }
#pragma unroll
                        for (int uu = 0; uu < RBF; ++uu) {
                            const int slot = (u0 + uu) * G::THREADS + tid_x;
                            const int row = slot / 48, c = slot - row * 48;
                            const int m = em0 + hf * 96 + row, n = en0 + c * 4;
                            if (m >= p.M || n >= p.N) continue;
                            f32x4 v = *(const f32x4*)(smem + row * FSTRIDE + c * 16);
                            const int64_t orow = orow_of(hf * 96 + row);
                            if (p.residual) v += res[uu];
                            if (p.rowmod) v += *(const f32x4*)(p.rowmod + (int64_t)(m % p.rowmod_period) * p.N + n);
                            if (p.out_scale != 0.f) v *= p.out_scale;
                            *(f32x4*)((float*)p.out + orow * p.ldo + n) = v;
                            if (p.out2) st_stream((bf16x4*)((bf16_t*)p.out2 + orow * p.ldo2 + n), (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])});
                        }
                        if (u0 + RBF < NU) {
#pragma unroll
                            for (int u = 0; u < RBF; ++u) res[u] = nxt[u];
                        }
                    }
                    continue;
                }
#pragma unroll 1
                for (int u = 0; u < NU; ++u) {
                    const int slot = u * G::THREADS + tid_x;
                    const int row = slot / 48, c = slot - row * 48;
                    const int m = em0 + hf * 96 + row, n = en0 + c * 4;
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
            continue;
        }
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int m = em0 + wm * 96 + i * 16 + fr_x;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = en0 + wn * 48 + j * 16 + fq_x * 4;
            if (n >= p.N) continue;
            nt_epilogue<EPI>(p, omap, m, n, acc[i][j]);
        }
    }
    }   // tile loop
}

// (Round 5 built two more NT kernels on this tile and removed them again after measuring them -- both bit-identical to gemm_nt192_kernel, neither
// faster: gemm_nt192d_kernel, K = 768 with the epilogue of output tile i inside the K loop of tile i + 1 (k-step register pipeline, a ring that never
// drains, run-time counted vmcnt, LDS-staged 16-row blocks), and gemm_nt192w4_kernel, four waves of 96 x 96 outputs with the accumulators pinned in
// AGPRs.  Source: git commit 6dc179d; records: profiles/r05_gemm_deferred_epilogue*.log, r05_gemm_four_wave_prototype.log; DESIGN.md section 5 "Round 5".)

// ------------------------------------------------------------------------------------------------ TN
struct TN192Args {
    vtGemmTN p[VT_TN_MAX_GROUP];
    int tile_start[VT_TN_MAX_GROUP + 1];
    int n;
};

// [64 m-rows][192 cols] bf16 image, 384-B rows = 24 chunks; physical chunk = (lc & ~7) | ((lc & 7) ^ f(row))
__device__ __forceinline__ int swz_tn192(int row) { return (((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1; }

// lane byte offsets of the 3 DMA pieces inside a [64 m-rows][ncols] slab starting at column c0 (invariant over K-tiles)
__device__ __forceinline__ void tn192_piece_offsets(int64_t ld, int c0, int ncols, int tid, unsigned (&off)[3]) {
    const int maxchunk = (ncols >> 3) - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int slot = i * 512 + tid;
        const int row = slot / 24;
        const int pc = slot - row * 24;
        const int lc = (pc & ~7) | ((pc & 7) ^ swz_tn192(row));
        int gc = (c0 >> 3) + lc;
        gc = gc < maxchunk ? gc : maxchunk;
        off[i] = (unsigned)((row * ld + gc * 8) * 2);
    }
}
__device__ __forceinline__ void stage_tn192(const bf16_t* slab_row0, const unsigned (&off)[3], unsigned lds, int wave) {
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16_sv(slab_row0, off[i], lds + (i * 512 + wave * 64) * 16);
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
    extern __shared__ __attribute__((aligned(128))) char smem[];
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
    // tiles walk in groups of 4 p-rows, p fastest: an XCD's round of 32 consecutive tiles is a 4 x 8 (or 8 x 4) block of the output,
    // i.e. 12 operand panels instead of the 18 a row-major walk gives a wide problem (fc2's 4 x 16 tiles: 2 x 16 per round)
    const int tiles_p = (p.p_lim + TM - 1) / TM;
    const int grp4 = local / (4 * tiles_q), rem4 = local - grp4 * 4 * tiles_q;
    const int rows4 = min(4, tiles_p - grp4 * 4);
    const int p0 = (grp4 * 4 + rem4 % rows4) * TM, q0 = (rem4 / rows4) * TN_;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.M / TK;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    unsigned aoff[3], boff[3];
    tn192_piece_offsets(p.lda, p0, p.P, tid, aoff);
    tn192_piece_offsets(p.ldb, q0, p.Q, tid, boff);
    stage_tn192(A, aoff, sbase, wave);
    stage_tn192(B, boff, sbase + OP_BYTES, wave);
    if (nt > 1) {
        stage_tn192(A + (int64_t)TK * p.lda, aoff, sbase + STAGE_BYTES, wave);
        stage_tn192(B + (int64_t)TK * p.ldb, boff, sbase + STAGE_BYTES + OP_BYTES, wave);
    }
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) wait_vmcnt6(); else wait_vmcnt0();
        raw_barrier();
        if (t + 2 < nt) {
            int nx = cur + 2; nx = nx >= NSTAGE ? nx - NSTAGE : nx;
            stage_tn192(A + (int64_t)(t + 2) * TK * p.lda, aoff, sbase + nx * STAGE_BYTES, wave);
            stage_tn192(B + (int64_t)(t + 2) * TK * p.ldb, boff, sbase + nx * STAGE_BYTES + OP_BYTES, wave);
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


// ------------------------------------------------------------------------------------------------ TN, register-pipelined
// Round 4: the weight-gradient kernel with the treatment the NT kernel got in round 1.  The kernel above reads a K-tile's 36
// transposed fragments right behind the barrier that publishes it, so after every barrier all 8 waves burst-read the LDS
// before the first MFMA can issue (the NT ablation of that pattern: 12 % of the kernel).  Here tile t is multiplied out of one
// REGISTER set while the 36 reads of tile t+1 go into the other set, three between every row of three MFMAs, and the 6 LDS-DMA
// pieces of tile t+3 are issued in the middle of the body into the LDS slot tile t vacated one barrier earlier.  All reads are
// inline asm with explicit waits (one lgkmcnt(0) at the end of a body), DMAs are ordered by a counted vmcnt(6) + raw barrier.
// Same MFMA order per accumulator as the kernel above (k-half 0 then 1 of every K-tile, tiles ascending): bit-identical output.
// A fragment's two reads differ by an immediate (rows +4 = +1536 B, swizzle unchanged: f(row) reads row bits 1 and 3), the
// k-half by another (+32 rows = +12288 B), so a wave carries 9 per-lane base offsets (6 A + 3 B column groups).
__global__ __launch_bounds__(512, 2) void gemm_tn192p_kernel(const TN192Args a) {
    extern __shared__ __attribute__((aligned(128))) char smem[];
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
    // tiles walk in groups of 4 p-rows, p fastest: an XCD's round of 32 consecutive tiles is a 4 x 8 (or 8 x 4) block of the output,
    // i.e. 12 operand panels instead of the 18 a row-major walk gives a wide problem (fc2's 4 x 16 tiles: 2 x 16 per round)
    const int tiles_p = (p.p_lim + TM - 1) / TM;
    const int grp4 = local / (4 * tiles_q), rem4 = local - grp4 * 4 * tiles_q;
    const int rows4 = min(4, tiles_p - grp4 * 4);
    const int p0 = (grp4 * 4 + rem4 % rows4) * TM, q0 = (rem4 / rows4) * TN_;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;

    f32x4 acc[6][3];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nt = p.M / TK;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    unsigned aoff[3], boff[3];
    tn192_piece_offsets(p.lda, p0, p.P, tid, aoff);
    tn192_piece_offsets(p.ldb, q0, p.Q, tid, boff);
    auto slot_of = [](int t) { return t % NSTAGE; };
    auto issue_tile = [&](int t) {
        const unsigned dst = sbase + slot_of(t) * STAGE_BYTES;
        stage_tn192(A + (int64_t)t * TK * p.lda, aoff, dst, wave);
        stage_tn192(B + (int64_t)t * TK * p.ldb, boff, dst + OP_BYTES, wave);
    };
    // per-lane byte offset of the first read (rows kb + 8g + q, k-half 0) of column group `col` inside an operand image
    auto frag_off = [&](int col) {
        const int gg = lane >> 4, lam = lane & 15;
        const int q = lam >> 2, pp = lam & 3;
        const int lc = (col >> 3) + (pp >> 1);
        const int r0 = 8 * gg + q;
        const int c0 = (lc & ~7) | ((lc & 7) ^ swz_tn192(r0));
        return (unsigned)(r0 * 384 + (c0 << 4) + ((pp & 1) << 3));
    };
    unsigned fa[6], fb[3];
#pragma unroll
    for (int i = 0; i < 6; ++i) fa[i] = frag_off(wm * 96 + i * 16);
#pragma unroll
    for (int j = 0; j < 3; ++j) fb[j] = OP_BYTES + frag_off(wn * 48 + j * 16);

    issue_tile(0);
    if (nt > 1) issue_tile(1);
    if (nt > 2) issue_tile(2);
    if (nt > 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (nt > 1) wait_vmcnt6();
    else wait_vmcnt0();
    raw_barrier();

    // fragment register sets: lo = rows r0.., hi = rows r0 + 4..; [k-half][column group]
    bf16x4 xal[2][6], xah[2][6], xbl[2][3], xbh[2][3], yal[2][6], yah[2][6], ybl[2][3], ybh[2][3];
#define VT_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define VT_TRF(lo, hi, addr, kh)                 \
    VT_TR(lo, addr, (kh) * 12288);               \
    VT_TR(hi, addr, (kh) * 12288 + 1536)
#define VT_MROW(i, kh, Cal, Cah, Cbl, Cbh)                                                                                           \
    {                                                                                                                                \
        const bf16x8 af_ = cat4(Cal[kh][i], Cah[kh][i]);                                                                             \
        acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat4(Cbl[kh][0], Cbh[kh][0]), af_, acc[i][0], 0, 0, 0);                   \
        acc[i][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat4(Cbl[kh][1], Cbh[kh][1]), af_, acc[i][1], 0, 0, 0);                   \
        acc[i][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat4(Cbl[kh][2], Cbh[kh][2]), af_, acc[i][2], 0, 0, 0);                   \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }
    // compute tile from set C while prefetching the tile at LDS address `nb` into set N; DMA of `dma_tile` in the middle
#define VT_TSTEP(Cal, Cah, Cbl, Cbh, Nal, Nah, Nbl, Nbh, nb)                                                                         \
    {                                                                                                                                \
        unsigned na_[6], nb_[3];                                                                                                     \
        _Pragma("unroll") for (int i_ = 0; i_ < 6; ++i_) na_[i_] = (nb) + fa[i_];                                                    \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) nb_[j_] = (nb) + fb[j_];                                                    \
        VT_TRF(Nbl[0][0], Nbh[0][0], nb_[0], 0); VT_TR(Nbl[0][1], nb_[1], 0);                                                        \
        VT_MROW(0, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nbh[0][1], nb_[1], 1536); VT_TRF(Nbl[0][2], Nbh[0][2], nb_[2], 0);                                                     \
        VT_MROW(1, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TRF(Nal[0][0], Nah[0][0], na_[0], 0); VT_TR(Nal[0][1], na_[1], 0);                                                        \
        VT_MROW(2, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nah[0][1], na_[1], 1536); VT_TRF(Nal[0][2], Nah[0][2], na_[2], 0);                                                     \
        VT_MROW(3, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TRF(Nal[0][3], Nah[0][3], na_[3], 0); VT_TR(Nal[0][4], na_[4], 0);                                                        \
        VT_MROW(4, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nah[0][4], na_[4], 1536); VT_TRF(Nal[0][5], Nah[0][5], na_[5], 0);                                                     \
        VT_MROW(5, 0, Cal, Cah, Cbl, Cbh);                                                                                           \
        if (dma_tile >= 0) {                                                                                                         \
            stage_tn192(A + (int64_t)dma_tile * TK * p.lda, aoff, dma_dst, wave);                                                    \
            stage_tn192(B + (int64_t)dma_tile * TK * p.ldb, boff, dma_dst + OP_BYTES, wave);                                         \
        }                                                                                                                            \
        VT_TRF(Nbl[1][0], Nbh[1][0], nb_[0], 1); VT_TR(Nbl[1][1], nb_[1], 12288);                                                    \
        VT_MROW(0, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nbh[1][1], nb_[1], 13824); VT_TRF(Nbl[1][2], Nbh[1][2], nb_[2], 1);                                                    \
        VT_MROW(1, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TRF(Nal[1][0], Nah[1][0], na_[0], 1); VT_TR(Nal[1][1], na_[1], 12288);                                                    \
        VT_MROW(2, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nah[1][1], na_[1], 13824); VT_TRF(Nal[1][2], Nah[1][2], na_[2], 1);                                                    \
        VT_MROW(3, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TRF(Nal[1][3], Nah[1][3], na_[3], 1); VT_TR(Nal[1][4], na_[4], 12288);                                                    \
        VT_MROW(4, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        VT_TR(Nah[1][4], na_[4], 13824); VT_TRF(Nal[1][5], Nah[1][5], na_[5], 1);                                                    \
        VT_MROW(5, 1, Cal, Cah, Cbl, Cbh);                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }
    {   // fragments of tile 0 (slot 0)
        unsigned na_[6], nb_[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) na_[i] = sbase + fa[i];
#pragma unroll
        for (int j = 0; j < 3; ++j) nb_[j] = sbase + fb[j];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (kh == 0) { VT_TRF(xbl[0][j], xbh[0][j], nb_[j], 0); } else { VT_TRF(xbl[1][j], xbh[1][j], nb_[j], 1); }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (kh == 0) { VT_TRF(xal[0][i], xah[0][i], na_[i], 0); } else { VT_TRF(xal[1][i], xah[1][i], na_[i], 1); }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    auto sync_for = [&](int nx) {   // tile nx landed for every wave; tile nx + 1 may stay in flight
        if (nx + 1 < nt) wait_vmcnt6(); else wait_vmcnt0();
        raw_barrier();
    };
    for (int t = 0;;) {
        {
            const bool more = t + 1 < nt;
            if (more) sync_for(t + 1);
            const unsigned nb = sbase + slot_of(more ? t + 1 : t) * STAGE_BYTES;
            const int dma_tile = t + NSTAGE < nt ? t + NSTAGE : -1;   // into the slot tile t vacated (its fragments are in registers)
            const unsigned dma_dst = sbase + slot_of(t) * STAGE_BYTES;
            VT_TSTEP(xal, xah, xbl, xbh, yal, yah, ybl, ybh, nb)
            if (++t == nt) break;
        }
        {
            const bool more = t + 1 < nt;
            if (more) sync_for(t + 1);
            const unsigned nb = sbase + slot_of(more ? t + 1 : t) * STAGE_BYTES;
            const int dma_tile = t + NSTAGE < nt ? t + NSTAGE : -1;
            const unsigned dma_dst = sbase + slot_of(t) * STAGE_BYTES;
            VT_TSTEP(yal, yah, ybl, ybh, xal, xah, xbl, xbh, nb)
            if (++t == nt) break;
        }
    }
#undef VT_TR
#undef VT_TRF
#undef VT_MROW
#undef VT_TSTEP

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

static int g_num_cus = 256;   // set by vt_gemm192_init from the device properties
static int g_order_env = -1;   // VT_GEMM_TILE_ORDER (read once by vt_gemm192_init): A/B timing of the tile order inside a whole step; -1 = automatic

// Tile order of a launch.  The tile list is dealt to the 8 XCDs in contiguous chunks (xcd_remap); with a row-major list the ~32 tiles in
// flight on one XCD are 32 / tiles_n whole tile rows, i.e. 32 / tiles_n A panels + ALL tiles_n B panels stream through its 4-MB L2 per round
// (qkv forward: 2.7 + 12 panels of 295 KB = 4.3 MB, re-fetched every round: FETCH_SIZE 4.4 x the operands).  In column blocks of W tile
// columns the chunk is a (chunk / W) x W rectangle: 32 / W + W panels per round, least near W = 6, and the W weight panels stay resident
// from round to round.  Staging alone (tools/probes/fill_probe.hip, profiles/r05_operand_staging_tile_order.log): 19.0 -> 16.6 us and
// 72 -> 38 MB fetched per pass on qkv forward.  Narrow outputs (tiles_n <= 4) keep the row-major list: every B panel is shared by all rows already.
// The width is vt_auto_col_block(tile columns, 32 tiles in flight per XCD): 6 for qkv forward (12 columns) and for N = 3072 (16 columns: 6 + 6 + 4;
// 6 and 8 measure the same there), row-major for the 4 tile columns of N = 768.

template <int WN>
static int nt192_in_flight(int ntiles) {   // tiles one XCD works on at a time
    const int per_xcd = (WN == 4 ? 1 : 2) * g_num_cus / 8, chunk = (ntiles + 7) / 8;
    return per_xcd < chunk ? per_xcd : chunk;
}

// Called by vt_gemm_nt / vt_gemm_tn_grouped (vt_gemm.hip) after argument validation.
template <int WN>
static void launch_nt192(const vtGemmNT& p, hipStream_t s, int dbg, int one_tile, int order) {
    using G = NTGeo<WN>;
    NT192Args a;
    a.p = p;
    a.dbg = dbg;
    a.tiles_m = (p.M + TM - 1) / TM;
    a.tiles_n = (p.N + G::TNW - 1) / G::TNW;
    if (order < 0) order = g_order_env;
    a.col_block = order >= 0 ? (order < a.tiles_n ? order : 0) : vt_auto_col_block(a.tiles_n, nt192_in_flight<WN>(a.tiles_m * a.tiles_n));
    // WN == 4: persistent, one workgroup per CU walks tiles b, b + grid, ...; one_tile (vtGemmNT.tile = 6, the data-parallel backward): one
    // tile per workgroup, so that the hardware dispatcher hands tiles to whichever CU is free while a collective's workgroups hold some.
    // A launch mode, not a timing ablation: `dbg` stays 0 and every epilogue keeps its production store path
    const int ntiles = a.tiles_m * a.tiles_n;
    const int persist = (WN == 4 && !one_tile) ? g_num_cus : ntiles;
    const dim3 grid(ntiles < persist ? ntiles : persist), block(G::THREADS);
    const size_t lds = G::NST * G::STAGE;
    constexpr size_t tab_gelu = WN == 4 ? GeluTab<VT_EPI_BF16_GELU>::BYTES : 0;
    switch (p.epi) {
        case VT_EPI_BF16: hipLaunchKernelGGL((gemm_nt192_kernel<VT_EPI_BF16, WN>), grid, block, lds, s, a); break;
        case VT_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_nt192_kernel<VT_EPI_BF16_GELU, WN>), grid, block, lds + tab_gelu, s, a); break;
        case VT_EPI_F32: hipLaunchKernelGGL((gemm_nt192_kernel<VT_EPI_F32, WN>), grid, block, lds, s, a); break;
        default: hipLaunchKernelGGL((gemm_nt192_kernel<VT_EPI_BF16_DGELU, WN>), grid, block, lds, s, a); break;
    }
}

// half == 0: 192x192 tiles, one workgroup per CU; half != 0: 192x96 tiles, two per CU
int vt_gemm_nt192_launch(const vtGemmNT& p, hipStream_t s, int dbg, int half, int one_tile, int order) {
    if (half) launch_nt192<2>(p, s, dbg, one_tile, order);
    else launch_nt192<4>(p, s, dbg, one_tile, order);
    return 0;
}

int vt_gemm_tn192_launch(const vtGemmTN* ph, int n, hipStream_t s, int burst) {
    TN192Args a;
    a.n = n;
    a.tile_start[0] = 0;
    for (int g = 0; g < n; ++g) {
        a.p[g] = ph[g];
        const int tp = (ph[g].p_lim + TM - 1) / TM, tq = (ph[g].q_lim + TN_ - 1) / TN_;
        a.tile_start[g + 1] = a.tile_start[g] + tp * tq;
    }
    // burst != 0 (vtGemmTN.tile = 7): the round-1 kernel that reads a tile's fragments behind its barrier, kept for A/B timing
    if (burst) hipLaunchKernelGGL(gemm_tn192_kernel, dim3(a.tile_start[n]), dim3(512), NSTAGE * STAGE_BYTES, s, a);
    else hipLaunchKernelGGL(gemm_tn192p_kernel, dim3(a.tile_start[n]), dim3(512), NSTAGE * STAGE_BYTES, s, a);
    return 0;
}

template <int EPI, int WN>
static hipError_t allow_lds_nt() {
    return hipFuncSetAttribute((const void*)gemm_nt192_kernel<EPI, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                               NTGeo<WN>::NST * NTGeo<WN>::STAGE + (WN == 4 ? GeluTab<EPI>::BYTES : 0));
}

#ifdef VT_GEMM_STAMPS
extern "C" int vt_gemm_nt_stamps(unsigned long long* host_out) {   // diagnostic build only: 256 x 8 x 8 counters
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nt_stamps), sizeof(g_nt_stamps)) == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
#endif

int vt_gemm192_init() {
    // more dynamic LDS than the 64 KiB default: opt in once per kernel
    static bool done = false;
    if (done) return 0;
    hipError_t e = hipSuccess;
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16, 4>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16_GELU, 4>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_F32, 4>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16_DGELU, 4>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16, 2>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16_GELU, 2>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_F32, 2>();
    if (e == hipSuccess) e = allow_lds_nt<VT_EPI_BF16_DGELU, 2>();
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn192p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
    if (e == hipSuccess) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_num_cus = cus;
        const char* ord = getenv("VT_GEMM_TILE_ORDER");
        if (ord && *ord) g_order_env = atoi(ord);
    }
    if (e != hipSuccess) {
        vt_set_error("vt_gemm192_init: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return VT_ERR_LAUNCH;
    }
    done = true;
    return 0;
}
