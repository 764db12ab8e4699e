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
        m0 = (sid / a.tiles_n) * TM, n0 = (sid % a.tiles_n) * G::TNW;
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
                            st_stream((bf16x4*)((bf16_t*)p.out + (int64_t)m * p.ldo + n), r);
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
                        st_stream_any((bf16x8*)((bf16_t*)p.out + (int64_t)m * p.ldo + n), h);
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
                            st_stream_any((bf16x8*)((bf16_t*)p.out2 + (int64_t)m * p.ldo2 + n), gl);
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
                    st_stream((bf16x4*)o, h);
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
                        st_stream((bf16x4*)((bf16_t*)p.out2 + dbg_shift + (int64_t)m * p.ldo2 + n), gl);
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

// ------------------------------------------------------------------------------------------------ NT, deferred epilogue
// gemm_nt192d_kernel: the 192x192x64 kernel above for the step's K = 768 launches with several output tiles per workgroup (qkv
// forward, fc1 forward + GELU, fc2 input gradient x gelu'): the epilogue of output tile i runs INSIDE the K loop of tile i + 1.
//
// Why: a third to a half of a K = 768 tile's cycles lay outside its main loop (stamps, DESIGN section 5: accumulators -> LDS image,
// read-back, GELU / gelu' arithmetic, stores, the ring drained and refilled around it), with the matrix pipe idle and every CU in
// the same phase.  The staged image (74 KB) cannot stay in the LDS next to the ring (144 KB) and a second accumulator set does not
// fit the registers -- but the OUTPUT does: 72 fp32 accumulators leave as 36 registers of bf16 (bias added, rounded: `held`), and
// what freed them is a register pipeline over k-STEPS (two 36-register fragment sets, one per 32-wide half of a K-tile, instead of
// two 72-register sets for whole K-tiles).  Per K-tile body: the first half's MFMAs run while the second half's fragments are
// read, then the barrier that publishes the next K-tile, then the second half's MFMAs with the next K-tile's first fragments and
// the LDS-DMA of K-tile + 3 into the slot just vacated.  The finished tile leaves one twelfth per body: a 16-row x 192-column block
// (rows i*16.. of wave row b & 1) is written by its four owner waves into one of two 6-KB staging images behind the ring in the
// first half, the body's barrier publishes it, and in the second half 384 threads read it back row-major -- 16 bytes each, a wave
// instruction = 1 KB of whole lines -- apply GELU (arithmetic: the look-up table does not fit next to two staging images) and
// store with streaming stores.  (First version: 8-byte stores straight from the MFMA layout, 16 rows x 32 bytes per instruction.
// Correct and SLOWER than the kernel above -- 57.0 / 85.5 us on qkv forward / fc1 GELU against 51.4 / 76.8 -- although the same K
// loop without any epilogue ran in 47.1 / 62.4: partial-line stores cost far more than their bytes; as streaming stores 69.5 / 188.)
// gelu' is applied by the owner waves before staging (their lanes' own rows: the column sums for the fc1 bias gradient stay a
// per-lane running sum, reduced once per tile), with the saved pre-activations loaded two bodies ahead in the MFMA layout.
// The ring never drains between output tiles: K-tiles 0..2 of the next tile are requested in bodies 9..11 of the current one.
// Same MFMA order per accumulator as the kernels above: bit-identical outputs.
//
// vmcnt is counted at RUN time: `issued` counts this wave's vector-memory instructions (LDS-DMA pieces, epilogue stores, bias
// and pre-activation loads), `mark[slot]` remembers its value behind the DMA of the K-tile in that slot, and the wait in front of a
// barrier is s_waitcnt vmcnt(issued - mark) through a scalar jump table -- no hand-derived immediates to get wrong when an epilogue
// changes what it issues.  (A smaller count than necessary only waits longer; the table clamps at 20.)
__device__ __forceinline__ void vm_wait_dyn(int n) {   // s_waitcnt vmcnt(min(n, 20)), n wave-uniform
#define VT_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n < 0 ? 0 : n > 20 ? 20 : n) {
        VT_VMW(0) VT_VMW(1) VT_VMW(2) VT_VMW(3) VT_VMW(4) VT_VMW(5) VT_VMW(6) VT_VMW(7) VT_VMW(8) VT_VMW(9) VT_VMW(10)
        VT_VMW(11) VT_VMW(12) VT_VMW(13) VT_VMW(14) VT_VMW(15) VT_VMW(16) VT_VMW(17) VT_VMW(18) VT_VMW(19) VT_VMW(20)
    }
#undef VT_VMW
}

// vector-memory accesses with a wave-uniform 64-bit base + a per-lane 32-bit byte offset, invisible to the compiler's wait counting:
// the caller counts them (`issued`) and waits by hand
// timing ablations of the deferred epilogue (A/B builds, WRONG results except VT_D_PLAINST): -DVT_D_NOSTORE no global stores, -DVT_D_NOSTAGE no
// staging writes / read-back, -DVT_D_PLAINST ordinary instead of streaming stores
__device__ __forceinline__ void st16_sv_nt(const void* base_uniform, unsigned lane_off, bf16x8 v) {
#ifdef VT_D_NOSTORE
    asm volatile("" ::"v"(lane_off), "v"(v), "s"(base_uniform));
    return;
#endif
#ifdef VT_D_PLAINST
    asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_off), "v"(v), "s"(base_uniform) : "memory");
    return;
#endif
    // (s_nop 1: a VALU write to the data registers of a 16-byte store needs wait states the compiler does not add behind inline asm)
    asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_off), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void ld8_sv(bf16x4& dst, const void* base_uniform, unsigned lane_off) {
    asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void ld16_sv(f32x4& dst, const void* base_uniform, unsigned lane_off) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(lane_off), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void st4_sv(const void* base_uniform, unsigned lane_off, float v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(lane_off), "v"(v), "s"(base_uniform) : "memory");
}
// sum over the 16 lanes of a DPP row (fixed tree order); every lane of the row ends with the total
__device__ __forceinline__ float row16_sum(float x) {
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));   // row_half_mirror
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));   // row_mirror
    return x;
}

// -DVT_D_NOUNITS (timing ablation, WRONG results): the K loop without the finished tile's epilogue
#ifdef VT_D_NOUNITS
#define VT_D_UNITS false
#else
#define VT_D_UNITS true
#endif
constexpr int NT192D_K = 768;                  // the only contraction length this kernel is built for (12 K-tiles per output tile)
constexpr int NT192D_SROW = TN_ * 2 + 8;       // staging image row: 192 bf16 + 8 bytes (the 16 rows of a ds_write_b64 group fall on 16 bank pairs)
constexpr int NT192D_SIMG = 16 * NT192D_SROW;  // one 16-row block
constexpr int NT192D_EXTRA = 2 * NT192D_SIMG + 4 * TN_ * 4;   // two staging images + (DGELU) the two wave rows' column sums, one set per tile parity

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt192d_kernel(const NT192Args a) {
    using G = NTGeo<4>;
    constexpr int NT = NT192D_K / TK;     // 12 K-tiles = 12 bodies = the 12 (wave row, 16-row) blocks of the previous tile
    static_assert(NT % G::NST == 0 && NT == 12, "a K-tile's ring slot must not depend on the output tile; one block per body");
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const vtGemmNT& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nwg = a.tiles_m * a.tiles_n;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wm * 96 + fr, brow = wn * 48 + fr;
    const unsigned a_off0 = arow * 128 + ((fq ^ ((arow >> 1) & 7)) << 4);
    const unsigned b_off0 = G::OPA + brow * 128 + ((fq ^ ((brow >> 1) & 7)) << 4);
    unsigned poff[G::P];                 // every tile of this kernel lies inside the matrix: one set of lane offsets
    nt192_piece_offsets<4>(p.lda, 0, TM, p.ldb, 0, TN_, tid, poff);
    char* const stage = smem + G::NST * G::STAGE;                                   // [2][16][SROW]
    [[maybe_unused]] float* const red0 = (float*)(stage + 2 * NT192D_SIMG);         // DGELU: [2 tile parities][2 wave rows][192] column sums
    [[maybe_unused]] int ppar = 0;        // parity of the finished tile: its sums are read behind body 11's barrier while no barrier separates that from the next tile's first sums

    int issued = 0, mark0 = 0, mark1 = 0, mark2 = 0;
    auto issue_tile = [&](const bf16_t* ap, const bf16_t* bp, int kt, int slot) {
        const unsigned dst = sbase + slot * G::STAGE;
#pragma unroll
        for (int k = 0; k < G::P; ++k) nt192_stage_piece<4>(ap + kt * TK, bp + kt * TK, poff, dst, k, wave);
        issued += G::P;
        if (slot == 0) mark0 = issued;
        else if (slot == 1) mark1 = issued;
        else mark2 = issued;
    };
    auto tile_coords = [&](int it, int& m0, int& n0) {
        const int sid = xcd_remap(it, nwg);
        m0 = (sid / a.tiles_n) * TM, n0 = (sid % a.tiles_n) * TN_;
    };

    // owner side of a block (MFMA layout): lane (fr, fq) of wave (wm, wn) holds row fr, columns wn*48 + j*16 + fq*4 .. + 3 of the block
    const unsigned st_off = (unsigned)(fr * NT192D_SROW + (wn * 48 + fq * 4) * 2);                        // + j * 32 bytes
    [[maybe_unused]] const unsigned voff_x = (unsigned)((fr * p.ldaux + wn * 48 + fq * 4) * 2);           // + j * 32 bytes (the saved pre-activations)
    const unsigned voff_b = (unsigned)((wn * 48 + fq * 4) * 4);
    // reader side: thread t < 384 takes 16 bytes: row t / 24, columns (t % 24) * 8 .. + 7 of the block
    const int rd_row = tid / 24, rd_c = tid - rd_row * 24;
    const bool reader = wave < 6;
    const unsigned rd_off = (unsigned)(rd_row * NT192D_SROW + rd_c * 16);
    const unsigned rd_o = (unsigned)((rd_row * p.ldo + rd_c * 8) * 2);
    [[maybe_unused]] const unsigned rd_o2 = (unsigned)((rd_row * p.ldo2 + rd_c * 8) * 2);

    // ---- state of the deferred epilogue
    bf16x4 held[18];                      // the finished tile: accumulator (i, j) at 3 i + j; shifted down three per owned block
    [[maybe_unused]] bf16x4 uq[3];        // DGELU: the pre-activations of this wave's next block, requested two bodies ahead
    [[maybe_unused]] int marku = 0;
    int pm0 = 0, pn0 = 0;                 // the finished tile's origin
    bool proc = false;                    // a finished tile is waiting in `held`
    bool red_pending = false;             // DGELU: both wave rows' column sums sit in `red`, to be added behind the next barrier
    int rm0 = 0, rn0 = 0, rpar = 0;

    // block `blk` (0..11: wave row blk & 1, rows (blk >> 1) * 16 .. + 15 of that wave row) of the finished tile
    auto block_row0 = [&](int blk) { return (blk & 1) * 96 + (blk >> 1) * 16; };
    // DGELU: request the pre-activations of block `blk` of the tile at (tm0, tn0) (owner waves)
    auto request_u = [&](int tm0, int tn0, int blk) {
        if constexpr (EPI == VT_EPI_BF16_DGELU) {
            const char* base = (const char*)p.aux + ((int64_t)(tm0 + block_row0(blk)) * p.ldaux + tn0) * 2;
#pragma unroll
            for (int j = 0; j < 3; ++j) ld8_sv(uq[j], base + j * 32, voff_x);
            issued += 3;
            marku = issued;
        }
    };
    // The finished tile's block b leaves in SLICES placed between the MFMA rows of body b (a block of epilogue instructions in one gap
    // stalls the matrix pipe: both waves of a SIMD run the same code in step).
    // Owner waves, first half, slices 0..5: three accumulators (held[0..2]) -> staging image `buf`; gelu' applied on the way (DGELU).
    auto own_step = [&](int sl, int buf, [[maybe_unused]] bool first_block) {
        char* const img = stage + buf * NT192D_SIMG;
        if constexpr (EPI == VT_EPI_BF16_DGELU) {
            // twelve values, two per slice, one at a time (scheduling fences: interleaved evaluations keep ~25 temporaries alive)
            const int j = sl >> 1, e0 = (sl & 1) * 2;
#pragma unroll
            for (int e = e0; e < e0 + 2; ++e) {
                held[j][e] = f2bf(bf2f(held[j][e]) * gelu_erf_grad(bf2f(uq[j][e])));
                __builtin_amdgcn_sched_barrier(0);
            }
            if (sl & 1) {
                *(bf16x4*)(img + st_off + j * 32) = held[j];
                // column sums of the rounded values (the fc1 bias gradient): the block's 16 rows by a fixed DPP tree, the wave row's six
                // blocks one after the other in `red` (the same lanes own the same columns in every block: no other writer)
                if (p.colsum_partial) {
                    f32x4 v = {row16_sum(bf2f(held[j][0])), row16_sum(bf2f(held[j][1])), row16_sum(bf2f(held[j][2])), row16_sum(bf2f(held[j][3]))};
                    if (fr == 0) {
                        f32x4* dst = (f32x4*)(red0 + ppar * 2 * TN_ + wm * TN_ + wn * 48 + j * 16 + fq * 4);
                        if (!first_block) v += *dst;
                        *dst = v;
                    }
                }
            }
        } else {
#ifndef VT_D_NOSTAGE
            if (sl >= 1 && sl <= 3) *(bf16x4*)(img + st_off + (sl - 1) * 32) = held[sl - 1];
#endif
        }
    };
    auto own_done = [&]() {
#pragma unroll
        for (int q = 0; q < 15; ++q) held[q] = held[q + 3];
    };
    // Reader threads, second half, slices 0..5: 16 bytes of block `blk` out of staging image `buf` -> GELU (arithmetic, in place) -> global memory
    bf16x8 rh;
    auto rd_step = [&](int sl, int blk, int buf) {
        if (!reader) return;
        if (sl == 0) {
#ifndef VT_D_NOSTAGE
            const char* src = stage + buf * NT192D_SIMG + rd_off;
            rh = cat4(*(const bf16x4*)src, *(const bf16x4*)(src + 8));
#else
            asm volatile("" : "=v"(rh));
#endif
        }
        const int64_t row0 = pm0 + block_row0(blk);
        if (sl == 1) {
            st16_sv_nt((const char*)p.out + (row0 * p.ldo + pn0) * 2, rd_o, rh);
            issued += 1;
        }
        if constexpr (EPI == VT_EPI_BF16_GELU) {
            if (sl >= 1 && sl <= 4) {
                const int e0 = (sl - 1) * 2;
#pragma unroll
                for (int e = e0; e < e0 + 2; ++e) {
                    rh[e] = f2bf(gelu_erf(bf2f(rh[e])));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (sl == 5) {
                st16_sv_nt((const char*)p.out2 + (row0 * p.ldo2 + pn0) * 2, rd_o2, rh);
                issued += 1;
            }
        }
    };
    auto cs_finish = [&]() {      // behind a barrier that follows both wave rows' cs_to_lds
        if constexpr (EPI == VT_EPI_BF16_DGELU) {
            if (red_pending) {
                if (wave < 3) {   // tid < 192: one column each, wave row 0 + wave row 1
                    const float v = red0[rpar * 2 * TN_ + tid] + red0[rpar * 2 * TN_ + TN_ + tid];
                    st4_sv(p.colsum_partial + (int64_t)(rm0 / TM) * p.N + rn0, (unsigned)tid * 4u, v);
                    issued += 1;
                }
                red_pending = false;
            }
        }
    };

    // ---- prologue: K-tiles 0..2 of the first output tile
    int m0, n0;
    tile_coords(blockIdx.x, m0, n0);
    const bf16_t* Ap = A + (int64_t)m0 * p.lda;
    const bf16_t* Bp = B + (int64_t)n0 * p.ldb;
    issue_tile(Ap, Bp, 0, 0);
    issue_tile(Ap, Bp, 1, 1);
    issue_tile(Ap, Bp, 2, 2);
    vm_wait_dyn(issued - mark0);
    raw_barrier();
    bf16x8 xa[6], xb[3], ya[6], yb[3];    // fragments of k-step 0 (x) / k-step 1 (y) of a K-tile
#define VT_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
#define VT_ROW(accrow, bb, aa)                                                                   \
    accrow[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[0], aa, accrow[0], 0, 0, 0);          \
    accrow[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[1], aa, accrow[1], 0, 0, 0);          \
    accrow[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb[2], aa, accrow[2], 0, 0, 0);          \
    __builtin_amdgcn_sched_barrier(0)
    {
        const unsigned na0 = sbase + a_off0, nb0 = sbase + b_off0;
        VT_DSR(xb[0], nb0, 0); VT_DSR(xb[1], nb0, 2048); VT_DSR(xb[2], nb0, 4096);
        VT_DSR(xa[0], na0, 0); VT_DSR(xa[1], na0, 2048); VT_DSR(xa[2], na0, 4096);
        VT_DSR(xa[3], na0, 6144); VT_DSR(xa[4], na0, 8192); VT_DSR(xa[5], na0, 10240);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }

    // One K-tile body; R = b % 3 is static: the ring slot of K-tile b.  The six LDS-DMA pieces of K-tile b + 3 go out one per MFMA row:
    // three in this body's second half (the slot is free behind the barrier), three in the next body's first half.
    const bf16_t *dpa = A, *dpb = B;      // the K-tile whose last three pieces are still to be issued
    unsigned ddst = 0;
    bool dpend = false;
#define VT_DMA3(first)                                                                                                                 \
    _Pragma("unroll") for (int k_ = (first); k_ < (first) + 3; ++k_) nt192_stage_piece<4>(dpa, dpb, poff, ddst, k_, wave);
#define VT_DMA_A                                                                                                                       \
    if (dpend) { VT_DMA3(3) issued += 3; }                                                                                             \
    if (dpend) { if (PSLOT == 0) mark0 = issued; else if (PSLOT == 1) mark1 = issued; else mark2 = issued; }
#define VT_DMA_B0 if (dpend) { VT_DMA3(0) issued += 3; }
#define VT_DMA_B1
#define VT_DMA_B2

// The block's epilogue sits in ONE place per half body.  One slice per MFMA row instead (a scalar branch and a scheduling fence each) measured
// slower: 53.6 / 64.2 / 89.8 / 105.0 us against 52.6 / 62.8 / 84.1 / 88.5 on qkv forward / fc1 plain / fc1 GELU / fc2 gelu'
// (profiles/r05_gemm_deferred_epilogue.log); so did giving a body's twelve DMA pieces to one wave of every SIMD (54.7 against 51.3).
#define VT_OWN(sl) if ((sl) == 3 && VT_D_UNITS && proc && owner) { _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_) own_step(s_, b & 1, b < 2); }
#define VT_RD(sl) if ((sl) == 3 && VT_D_UNITS && proc) { _Pragma("unroll") for (int s_ = 0; s_ < 6; ++s_) rd_step(s_, b, b & 1); }
#define VT_BODY(R, bexpr)                                                                                                              \
    {                                                                                                                                  \
        const int b = (bexpr);                                                                                                         \
        constexpr int SLOT = (R), NSLOT = ((R) + 1) % 3, PSLOT = ((R) + 2) % 3;                                                        \
        const unsigned cur = sbase + SLOT * G::STAGE, nx = sbase + NSLOT * G::STAGE;                                                   \
        unsigned ao_ = a_off0, bo_ = b_off0;     /* opaque: the three bodies' fragment addresses are loop-invariant and would be hoisted (12 registers) */ \
        asm volatile("" : "+v"(ao_), "+v"(bo_));                                                                                       \
        const bool owner = wm == (b & 1);        /* this wave's rows are in block b of the finished tile */                            \
        if constexpr (EPI != VT_EPI_BF16_DGELU) {                                                                                      \
            if (b == NT - 3 && p.bias) {                                                                                               \
                _Pragma("unroll") for (int j = 0; j < 3; ++j) ld16_sv(b4[j], (const char*)(p.bias + n0 + j * 16), voff_b);             \
                issued += 3;                                                                                                           \
                markb = issued;                                                                                                        \
            }                                                                                                                          \
        } else {                                                                                                                       \
            if (VT_D_UNITS && proc && owner) {                                                                                         \
                vm_wait_dyn(issued - marku);                                                                                           \
                asm volatile("" : "+v"(uq[0]), "+v"(uq[1]), "+v"(uq[2]));                                                              \
            }                                                                                                                          \
        }                                                                                                                              \
        /* ---- first half: k-step 0 out of x while k-step 1 of this K-tile goes into y; the owner waves stage block b */              \
        {                                                                                                                              \
            const unsigned na1 = (cur + ao_) ^ 64u, nb1 = (cur + bo_) ^ 64u;                                                           \
            VT_DSR(yb[0], nb1, 0); VT_DSR(yb[1], nb1, 2048);                                                                           \
            VT_ROW(acc[0], xb, xa[0]);                                                                                                 \
            VT_DSR(yb[2], nb1, 4096); VT_DSR(ya[0], na1, 0);                                                                           \
            VT_OWN(0);                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[1], xb, xa[1]);                                                                                                 \
            VT_DSR(ya[1], na1, 2048); VT_DSR(ya[2], na1, 4096);                                                                        \
            VT_DMA_A                                                                                                                   \
            VT_OWN(1);                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[2], xb, xa[2]);                                                                                                 \
            VT_DSR(ya[3], na1, 6144); VT_DSR(ya[4], na1, 8192);                                                                        \
            VT_OWN(2);                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[3], xb, xa[3]);                                                                                                 \
            VT_DSR(ya[5], na1, 10240);                                                                                                 \
            VT_OWN(3);                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[4], xb, xa[4]);                                                                                                 \
            VT_OWN(4);                                                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[5], xb, xa[5]);                                                                                                 \
            VT_OWN(5);                                                                                                                 \
            if (VT_D_UNITS && proc && owner) {                                                                                         \
                own_done();                                                                                                            \
            }                                                                                                                          \
            if constexpr (EPI == VT_EPI_BF16_DGELU) {                                                                                  \
                /* the pre-activations of this wave's next block (two bodies on): of the finished tile while b <= 9, of THIS tile in   \
                   its last two bodies (blocks 0 and 1 leave during the next tile's bodies 0 and 1, or first in the drain) */          \
                if (owner && (b >= NT - 2 || proc)) {                                                                                  \
                    if (b >= NT - 2) request_u(m0, n0, b - (NT - 2));                                                                  \
                    else request_u(pm0, pn0, b + 2);                                                                                   \
                }                                                                                                                      \
            }                                                                                                                          \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
        }                                                                                                                              \
        /* ---- the next K-tile (this tile's b + 1, or the next tile's 0) and the staged block become visible; this K-tile's slot is free */ \
        if (!(last && b == NT - 1)) vm_wait_dyn(issued - (NSLOT == 0 ? mark0 : NSLOT == 1 ? mark1 : mark2));                           \
        raw_barrier();                                                                                                                 \
        if (EPI == VT_EPI_BF16_DGELU && VT_D_UNITS && proc && b == NT - 1 && p.colsum_partial) { red_pending = true; rm0 = pm0; rn0 = pn0; rpar = ppar; }  \
        cs_finish();        /* (both wave rows wrote their sums in front of this barrier: wave row 0 in body 10, wave row 1 in body 11) */ \
        /* ---- second half: k-step 1 out of y, k-step 0 of the next K-tile into x, half the DMA of K-tile + 3, block b's read-back and stores */ \
        {                                                                                                                              \
            const unsigned na0 = nx + ao_, nb0 = nx + bo_;                                                                             \
            dpend = true;                                                                                                              \
            if (b + G::NST < NT) { dpa = Ap + (b + G::NST) * TK; dpb = Bp + (b + G::NST) * TK; }                                       \
            else if (!last) { dpa = Apn + (b + G::NST - NT) * TK; dpb = Bpn + (b + G::NST - NT) * TK; }                                \
            else dpend = false;                                                                                                        \
            ddst = cur;                                                                                                                \
            VT_DSR(xb[0], nb0, 0); VT_DSR(xb[1], nb0, 2048); VT_DSR(xb[2], nb0, 4096);                                                 \
            VT_RD(0);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[0], yb, ya[0]);                                                                                                 \
            VT_DSR(xa[0], na0, 0); VT_DSR(xa[1], na0, 2048); VT_DSR(xa[2], na0, 4096);                                                 \
            VT_RD(1);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[1], yb, ya[1]);                                                                                                 \
            VT_DSR(xa[3], na0, 6144); VT_DSR(xa[4], na0, 8192); VT_DSR(xa[5], na0, 10240);                                             \
            VT_RD(2);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[2], yb, ya[2]);                                                                                                 \
            VT_DMA_B0                                                                                                                  \
            VT_RD(3);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[3], yb, ya[3]);                                                                                                 \
            VT_DMA_B1                                                                                                                  \
            VT_RD(4);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[4], yb, ya[4]);                                                                                                 \
            VT_DMA_B2                                                                                                                  \
            VT_RD(5);                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            VT_ROW(acc[5], yb, ya[5]);                                                                                                 \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
        }                                                                                                                              \
    }

    for (int it = blockIdx.x; it < nwg; it += (int)gridDim.x) {
        const bool last = it + (int)gridDim.x >= nwg;
        int nm0 = 0, nn0 = 0;
        if (!last) tile_coords(it + gridDim.x, nm0, nn0);
        const bf16_t* Apn = A + (int64_t)nm0 * p.lda;
        const bf16_t* Bpn = B + (int64_t)nn0 * p.ldb;
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
        [[maybe_unused]] f32x4 b4[3];      // bias of this tile's columns, requested three bodies before its end
        [[maybe_unused]] int markb = 0;
#pragma unroll 1
        for (int b3 = 0; b3 < NT; b3 += 3) {
            VT_BODY(0, b3)
            VT_BODY(1, b3 + 1)
            VT_BODY(2, b3 + 2)
        }
        // ---- this tile is finished: its output waits in `held` (bias added, rounded) while the next tile's K loop runs
        if constexpr (EPI != VT_EPI_BF16_DGELU) {
            if (p.bias) {
                vm_wait_dyn(issued - markb);   // (the DMA of the last three bodies stays in flight)
#pragma unroll
                for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(b4[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) b4[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 v = acc[i][j];
                if constexpr (EPI != VT_EPI_BF16_DGELU) v += b4[j];
                held[i * 3 + j] = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
        pm0 = m0, pn0 = n0, proc = true;
        if constexpr (EPI == VT_EPI_BF16_DGELU) ppar ^= 1;
        m0 = nm0, n0 = nn0, Ap = Apn, Bp = Bpn;
    }
#undef VT_BODY
#undef VT_OWN
#undef VT_RD
#undef VT_DMA3
#undef VT_DMA_A
#undef VT_DMA_B0
#undef VT_DMA_B1
#undef VT_DMA_B2
#undef VT_DSR
#undef VT_ROW
    // ---- drain: the last tile's twelve blocks, with nothing left to hide them under (a barrier per block, two staging images in turn)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nothing of the ring is in flight any more: the drain counts nothing)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    cs_finish();
#pragma unroll 1
    for (int blk = 0; blk < NT; ++blk) {
        if (wm == (blk & 1)) {
            if constexpr (EPI == VT_EPI_BF16_DGELU) {
                // blocks 0 and 1 were requested in the last two bodies of the K loop; the later ones of this wave row go out one block ahead
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("" : "+v"(uq[0]), "+v"(uq[1]), "+v"(uq[2]));
            }
#pragma unroll
            for (int sl = 0; sl < 6; ++sl) own_step(sl, blk & 1, blk < 2);
            own_done();
            if constexpr (EPI == VT_EPI_BF16_DGELU) {
                if (blk + 2 < NT) request_u(pm0, pn0, blk + 2);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        raw_barrier();
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) rd_step(sl, blk, blk & 1);
    }
    if constexpr (EPI == VT_EPI_BF16_DGELU) {
        if (p.colsum_partial) { red_pending = true; rm0 = pm0; rn0 = pn0; rpar = ppar; }
        cs_finish();
    }
}

// ------------------------------------------------------------------------------------------------ NT, four waves (timing prototype)
// gemm_nt192w4_kernel (vtGemmNT.tile = 20, plain bf16 epilogue only): the same 192 x 192 x 64 tile and ring with FOUR waves, one per SIMD, each
// owning 96 x 96 outputs (6 x 6 accumulators = 144 registers of the 512 a lone wave may hold).  Why: round 5 found the K loop LDS-bound -- per
// K-tile and CU the LDS-DMA writes 48 KB (~870 array cycles at the ~56 B/clk that path sustains) and the eight 96 x 48 waves read 144 KB of
// fragments back (576 cycles): 1446 array cycles next to 1152 matrix-pipe cycles.  A 96 x 96 wave tile reads (96 + 96) x 128 B x 4 waves = 96 KB
// (384 cycles): 1254.  The price: a lone wave has no partner to cover its 12 DMA pieces and 24 fragment reads per K-tile; they are dealt two
// reads and two pieces per MFMA row.  Epilogue: 8-byte stores straight from the MFMA layout (a prototype for the main loop's speed; tile time over
// K gives the per-K-tile cost).
template <int EPI>
__global__ __launch_bounds__(256, 1) void gemm_nt192w4_kernel(const NT192Args a) {
    static_assert(EPI == VT_EPI_BF16, "prototype: plain bf16 epilogue");
    constexpr int OPA = TM * TK * 2, STAGE = 2 * OPA, NST = 3, P = 12;     // 12 DMA pieces per thread and K-tile (6 A + 6 B)
    extern __shared__ __attribute__((aligned(128))) char smem[];
    const vtGemmNT& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nwg = a.tiles_m * a.tiles_n;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* B = (const bf16_t*)p.B;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wm * 96 + fr, brow = wn * 96 + fr;
    const unsigned a_off0 = arow * 128 + ((fq ^ ((arow >> 1) & 7)) << 4);
    const unsigned b_off0 = OPA + brow * 128 + ((fq ^ ((brow >> 1) & 7)) << 4);
    unsigned poff[P];
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const int i = k % 6;
        const int slot = i * 256 + tid;
        const int row = slot >> 3;
        const int lc = (slot & 7) ^ ((row >> 1) & 7);
        poff[k] = (unsigned)((row * (k < 6 ? p.lda : p.ldb) + lc * 8) * 2);
    }
    const int nt = p.K / TK;
    auto piece = [&](const bf16_t* ap, const bf16_t* bp, unsigned dst, int k) {
        const int i = k % 6;
        glds16_sv(k < 6 ? ap : bp, poff[k], dst + (k < 6 ? 0 : OPA) + (i * 256 + wave * 64) * 16);
    };
#define VT_DSR(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off))
    // the MFMAs are inline asm with the accumulator pinned to the AGPR class ("+a"): as builtins hipcc moved 108 accumulator registers to the
    // VGPR file and back around every first half (216 v_accvgpr copies + 51 s_nop per K-tile)
#define VT_ROW6(accrow, bb, aa)                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < 6; ++j_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(accrow[j_]) : "v"(bb[j_]), "v"(aa)); \
    __builtin_amdgcn_sched_barrier(0)
    for (int it = blockIdx.x; it < nwg; it += (int)gridDim.x) {
        const int sid = xcd_remap(it, nwg);
        const int m0 = (sid / a.tiles_n) * TM, n0 = (sid % a.tiles_n) * TN_;
        const bf16_t* Ap = A + (int64_t)m0 * p.lda;
        const bf16_t* Bp = B + (int64_t)n0 * p.ldb;
        if (it != (int)blockIdx.x) __syncthreads();
#pragma unroll
        for (int t = 0; t < NST; ++t)
            if (t < nt) {
#pragma unroll
                for (int k = 0; k < P; ++k) piece(Ap + t * TK, Bp + t * TK, sbase + t * STAGE, k);
            }
        if (nt > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (nt > 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_barrier();
        f32x4 acc[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        bf16x8 xa[6], xb[6], ya[6], yb[6];
        {
            const unsigned na0 = sbase + a_off0, nb0 = sbase + b_off0;
            VT_DSR(xb[0], nb0, 0); VT_DSR(xb[1], nb0, 2048); VT_DSR(xb[2], nb0, 4096); VT_DSR(xb[3], nb0, 6144); VT_DSR(xb[4], nb0, 8192); VT_DSR(xb[5], nb0, 10240);
            VT_DSR(xa[0], na0, 0); VT_DSR(xa[1], na0, 2048); VT_DSR(xa[2], na0, 4096); VT_DSR(xa[3], na0, 6144); VT_DSR(xa[4], na0, 8192); VT_DSR(xa[5], na0, 10240);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
        int slot = 0;
#pragma unroll 1
        for (int t = 0; t < nt; ++t) {
            const int nslot = slot == 2 ? 0 : slot + 1;
            const unsigned cur = sbase + slot * STAGE, nx = sbase + nslot * STAGE;
            {   // first half: k-step 0 out of x, k-step 1 of this K-tile into y (two reads per MFMA row)
                const unsigned na1 = (cur + a_off0) ^ 64u, nb1 = (cur + b_off0) ^ 64u;
                VT_DSR(yb[0], nb1, 0); VT_DSR(yb[1], nb1, 2048);
                VT_ROW6(acc[0], xb, xa[0]);
                VT_DSR(yb[2], nb1, 4096); VT_DSR(yb[3], nb1, 6144);
                VT_ROW6(acc[1], xb, xa[1]);
                VT_DSR(yb[4], nb1, 8192); VT_DSR(yb[5], nb1, 10240);
                VT_ROW6(acc[2], xb, xa[2]);
                VT_DSR(ya[0], na1, 0); VT_DSR(ya[1], na1, 2048);
                VT_ROW6(acc[3], xb, xa[3]);
                VT_DSR(ya[2], na1, 4096); VT_DSR(ya[3], na1, 6144);
                VT_ROW6(acc[4], xb, xa[4]);
                VT_DSR(ya[4], na1, 8192); VT_DSR(ya[5], na1, 10240);
                VT_ROW6(acc[5], xb, xa[5]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if (t + 1 < nt) {
                if (t + 2 < nt) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            raw_barrier();
            {   // second half: k-step 1 out of y, k-step 0 of the next K-tile into x, the 12 DMA pieces of K-tile t + 3 two per MFMA row
                const unsigned na0 = nx + a_off0, nb0 = nx + b_off0;
                const bool dma = t + NST < nt;
                const bf16_t* ap = Ap + (t + NST) * TK;
                const bf16_t* bp = Bp + (t + NST) * TK;
                VT_DSR(xb[0], nb0, 0); VT_DSR(xb[1], nb0, 2048);
                if (dma) { piece(ap, bp, cur, 0); piece(ap, bp, cur, 1); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[0], yb, ya[0]);
                VT_DSR(xb[2], nb0, 4096); VT_DSR(xb[3], nb0, 6144);
                if (dma) { piece(ap, bp, cur, 2); piece(ap, bp, cur, 3); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[1], yb, ya[1]);
                VT_DSR(xb[4], nb0, 8192); VT_DSR(xb[5], nb0, 10240);
                if (dma) { piece(ap, bp, cur, 4); piece(ap, bp, cur, 5); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[2], yb, ya[2]);
                VT_DSR(xa[0], na0, 0); VT_DSR(xa[1], na0, 2048);
                if (dma) { piece(ap, bp, cur, 6); piece(ap, bp, cur, 7); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[3], yb, ya[3]);
                VT_DSR(xa[2], na0, 4096); VT_DSR(xa[3], na0, 6144);
                if (dma) { piece(ap, bp, cur, 8); piece(ap, bp, cur, 9); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[4], yb, ya[4]);
                VT_DSR(xa[4], na0, 8192); VT_DSR(xa[5], na0, 10240);
                if (dma) { piece(ap, bp, cur, 10); piece(ap, bp, cur, 11); }
                __builtin_amdgcn_sched_barrier(0);
                VT_ROW6(acc[5], yb, ya[5]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
        }
        // prototype epilogue: 8-byte stores from the MFMA layout (the MFMAs are asm: their results need the wait states hipcc would have added)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const int m = m0 + wm * 96 + i * 16 + fr, n = n0 + wn * 96 + j * 16 + fq * 4;
                f32x4 v = acc[i][j];
                if (p.bias) v += *(const f32x4*)(p.bias + n);
                *(bf16x4*)((bf16_t*)p.out + (int64_t)m * p.ldo + n) = (bf16x4){f2bf(v[0]), f2bf(v[1]), f2bf(v[2]), f2bf(v[3])};
            }
    }
#undef VT_DSR
#undef VT_ROW6
}

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

// Called by vt_gemm_nt / vt_gemm_tn_grouped (vt_gemm.hip) after argument validation.
template <int WN>
static void launch_nt192(const vtGemmNT& p, hipStream_t s, int dbg, int one_tile) {
    using G = NTGeo<WN>;
    NT192Args a;
    a.p = p;
    a.dbg = dbg;
    a.tiles_m = (p.M + TM - 1) / TM;
    a.tiles_n = (p.N + G::TNW - 1) / G::TNW;
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
int vt_gemm_nt192_launch(const vtGemmNT& p, hipStream_t s, int dbg, int half, int one_tile) {
    if (half) launch_nt192<2>(p, s, dbg, one_tile);
    else launch_nt192<4>(p, s, dbg, one_tile);
    return 0;
}

// The deferred-epilogue kernel (gemm_nt192d_kernel) takes a launch when every output tile lies inside the matrix, K = 768, the epilogue
// is one of the bf16 ones and the output rows are 8-byte aligned; vt_gemm_nt asks before it falls back to the kernels above.
bool vt_gemm_nt192d_eligible(const vtGemmNT& p) {
    return p.K == NT192D_K && p.M % TM == 0 && p.N % TN_ == 0 && p.epi != VT_EPI_F32 && p.ldo % 4 == 0 && ((uintptr_t)p.out & 7) == 0 &&
           (p.epi != VT_EPI_BF16_GELU || (p.ldo2 % 4 == 0 && ((uintptr_t)p.out2 & 7) == 0)) &&
           (p.epi != VT_EPI_BF16_DGELU || (p.ldaux % 4 == 0 && ((uintptr_t)p.aux & 7) == 0)) &&
           (int64_t)TM * p.ldo * 2 < (1ll << 31) && (int64_t)TM * (p.epi == VT_EPI_BF16_GELU ? p.ldo2 : p.epi == VT_EPI_BF16_DGELU ? p.ldaux : 8) * 2 < (1ll << 31);
}
int vt_gemm_nt192d_launch(const vtGemmNT& p, hipStream_t s, int one_tile) {
    NT192Args a;
    a.p = p;
    a.dbg = 0;
    a.tiles_m = p.M / TM;
    a.tiles_n = p.N / TN_;
    const int ntiles = a.tiles_m * a.tiles_n;
    const dim3 grid(one_tile ? ntiles : (ntiles < g_num_cus ? ntiles : g_num_cus)), block(512);
    const size_t ring = NTGeo<4>::NST * NTGeo<4>::STAGE;
    switch (p.epi) {
        case VT_EPI_BF16: hipLaunchKernelGGL((gemm_nt192d_kernel<VT_EPI_BF16>), grid, block, ring + NT192D_EXTRA, s, a); break;
        case VT_EPI_BF16_GELU: hipLaunchKernelGGL((gemm_nt192d_kernel<VT_EPI_BF16_GELU>), grid, block, ring + NT192D_EXTRA, s, a); break;
        default: hipLaunchKernelGGL((gemm_nt192d_kernel<VT_EPI_BF16_DGELU>), grid, block, ring + NT192D_EXTRA, s, a); break;
    }
    return 0;
}

int vt_gemm_nt192w4_launch(const vtGemmNT& p, hipStream_t s) {
    NT192Args a;
    a.p = p;
    a.dbg = 0;
    a.tiles_m = p.M / TM;
    a.tiles_n = p.N / TN_;
    const int ntiles = a.tiles_m * a.tiles_n;
    hipLaunchKernelGGL((gemm_nt192w4_kernel<VT_EPI_BF16>), dim3(ntiles < g_num_cus ? ntiles : g_num_cus), dim3(256), 3 * STAGE_BYTES, s, a);
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

int vt_gemm192_num_cus() { return g_num_cus; }

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
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192d_kernel<VT_EPI_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, NTGeo<4>::NST * NTGeo<4>::STAGE + NT192D_EXTRA);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192d_kernel<VT_EPI_BF16_GELU>, hipFuncAttributeMaxDynamicSharedMemorySize, NTGeo<4>::NST * NTGeo<4>::STAGE + NT192D_EXTRA);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192d_kernel<VT_EPI_BF16_DGELU>, hipFuncAttributeMaxDynamicSharedMemorySize, NTGeo<4>::NST * NTGeo<4>::STAGE + NT192D_EXTRA);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt192w4_kernel<VT_EPI_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn192_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_tn192p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE_BYTES);
    if (e == hipSuccess) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
            g_num_cus = cus;
    }
    if (e != hipSuccess) {
        vt_set_error("vt_gemm192_init: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return VT_ERR_LAUNCH;
    }
    done = true;
    return 0;
}
