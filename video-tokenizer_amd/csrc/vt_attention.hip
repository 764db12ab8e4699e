// Fused multi-head attention (no mask, head_dim 64 or 32) for gfx950: forward with online softmax and the
// two recompute backward kernels.  Replaces timm Attention's F.scaled_dot_product_attention inside
// timm Block (constructed at /root/reference/models/transformer.py:52-59) and its autograd.
//
// Layouts: qkv bf16 [B, L, 3, H, 64] (the qkv Linear's output, reshape (B,N,3,H,hd)); o / dO bf16
// [B, L, H, 64]; lse2 fp32 [B, H, L] = log2-domain log-sum-exp of the scaled scores; delta fp32 [B,H,L].
//
// Common skeleton (one workgroup = 4 waves, each wave owns 32 "stationary" rows held in registers as
// MFMA B operands; 64-row "streaming" tiles are staged global -> LDS with 16-B global_load_lds,
// double-buffered, one barrier per tile):
//   score-like products  X[stream_row][own_row] = Tile . Own^T      : A = tile rows by ds_read_b128
//   accumulate products  Acc^T[d][own_row]    += Tile^T . X         : A = tile columns by
//                        ds_read_b64_tr_b16; B = the fp32 accumulator X converted to bf16 IN REGISTERS
//                        (a 32x32 accumulator has its column on the lane and its rows in the 16
//                        registers, so registers 8s..8s+7 are the B fragment of k-step s; the k order
//                        inside a step is row 16s + 8(j>>2) + 4*half + (j&3), which the transposed read
//                        of the other operand follows).
// Because the own row (query in fwd/dQ, key in dK/dV) sits on the lane, softmax statistics are
// lane-local: no cross-lane reduction except one exchange between the two lane halves.
//
// LDS tile image: [64 rows][HD bf16].  head_dim 64: 128-B rows of eight 16-B chunks, physical chunk =
// logical ^ f(row), f(row) = (((row>>1)&1)<<2) | ((row>>2)&3); head_dim 32 (the GAN discriminator's heads,
// /root/reference/models/loss.py:119-204 with cfgs/larp_tokenizer.yaml:130-131): 64-B rows of four chunks,
// f(row) = (row>>2)&3.  Both are conflict-free for BOTH the b128 row reads of a 32x32x16 A operand and the
// transposed b64 reads (derivation in DESIGN.md).  The image is written lane-linearly by the LDS-DMA, so the
// XOR goes on the per-lane source address.
#include "vt_common.h"
#include <stdlib.h>

#include "vt_attn_tile.h"

// Diagnostic build only (-DVT_ATTN_STAMPS, tools/attn_stamps.sh): s_memtime stamps around the three segments of a tile iteration
// (tile body = staging issue + MFMA / softmax | s_waitcnt vmcnt(0) on the next tile's LDS-DMA | workgroup barrier), summed per wave.
// The shares are what to read, never the run time (cdna_hip_programming.md section 7, "In-kernel stamps").
#ifdef VT_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[3][2048][4][6];     // [kernel: fwd, dq, dkv][workgroup][wave][body, drain, barrier, iterations, realtime at loop start, at loop end (100 MHz)]
#define VT_STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[6] = {0, 0, 0, 0, __builtin_amdgcn_s_memrealtime(), 0}
#define VT_STAMP(i)                                                      \
    {                                                                    \
        const unsigned long long st_n = __builtin_amdgcn_s_memtime();    \
        st_acc[i] += st_n - st_t;                                        \
        st_t = st_n;                                                     \
    }
#define VT_STAMP_ITER st_acc[3] += 1
#define VT_STAMP_START st_t = __builtin_amdgcn_s_memtime()
#define VT_STAMP_FLUSH(kern)                                                                                       \
    st_acc[5] = __builtin_amdgcn_s_memrealtime();                                                                  \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048)                                                              \
        for (int i_ = 0; i_ < 6; ++i_) g_attn_stamps[kern][blockIdx.x][threadIdx.x >> 6][i_] = st_acc[i_]
#else
#define VT_STAMP_DECL
#define VT_STAMP(i)
#define VT_STAMP_ITER
#define VT_STAMP_START
#define VT_STAMP_FLUSH(kern)
#endif

namespace {

// one 64-key tile of the forward: S^T = K.Q^T, online softmax (log2 domain; max taken on the raw scores since
// the scale is positive), O^T += V^T.P^T.  TAIL masks keys >= L (last tile of a ragged sequence only).
// Round 4: fragments from precomputed lane offsets + immediates (TileAddr), and the softmax arithmetic two elements per
// instruction (v_pk_fma_f32 for s * c - m, v_pk_add_f32 for the row sum): 11.6 -> ~7 vector instructions per MFMA.
template <int HD, bool TAIL>
__device__ __forceinline__ void fwd_tile(unsigned kl, unsigned vl, const TileAddr<HD>& ad, const bf16x8 (&qf)[HD / 16], f32x16 (&oacc)[HD / 32],
                                         float& m, float& lsum, int key0, int L, float c, int half) {
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag_a<HD>(kl, ad, kt * 32, s), qf[s], sacc[kt], 0, 0, 0);
    }
    float mx = -__builtin_inff();
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (TAIL && (key0 + kt * 32 + reg_row(r, half) >= L)) sacc[kt][r] = -__builtin_inff();
            mx = fmaxf(mx, sacc[kt][r]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    // Lazy rescaling: `m` is the reference point of the exponentials, not necessarily the running maximum.  It moves (and
    // O, l are rescaled) only when some row's maximum has outgrown it by more than 2^8; until then p <= 256 instead of
    // <= 1, harmless in fp32 sums and in bf16 P (relative precision), and lse = m + log2(l) stays exact.  After the first
    // tiles the wave-uniform branch is almost never taken, which removes 16 packed multiplies + an exp per tile.
    const float want = mx * c;
    if (__builtin_amdgcn_ballot_w64(want > m + 8.0f) != 0ull) {
        const float mn = fmaxf(m, want);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);   // m = -inf on the first tile: alpha = 0, O and l are 0 anyway
        m = mn;
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < HD / 32; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
    }
    const f32x2 c2 = {c, c}, nm2 = {-m, -m};
    f32x2 ps2 = {0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const f32x2 sv = {sacc[kt][r], sacc[kt][r + 1]};
            const f32x2 t = __builtin_elementwise_fma(sv, c2, nm2);
            const f32x2 pv = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            ps2 += pv;
            sacc[kt][r] = pv[0];
            sacc[kt][r + 1] = pv[1];
        }
    lsum += ps2[0] + ps2[1];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc[kt], sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_a<HD>(vl, ad, kt * 32, sp, dt), pf, oacc[dt], 0, 0, 0);
        }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// CAUSAL (the AR consumer, models/larp_ar.py:186-190 `is_causal=True`): query q attends keys 0..q.  A workgroup stops at the last
// tile its 128 queries can see; per wave a tile is plain (every key <= every query of the wave), masked (fwd_tile<TAIL> with
// the per-lane limit q + 1 in place of L) or skipped.
template <int HD, bool CAUSAL = false>
__global__ __launch_bounds__(256, CAUSAL ? 3 : 4) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse2,
                                                           int L, int H, int nblk, float scale_log2e, int q_begin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    // 1-D grid, XCD-aware: the nblk row-blocks of one (batch, head) get consecutive ids inside ONE XCD's chunk, so the
    // K/V tiles they all stream are fetched into that XCD's L2 once instead of once per XCD
    const int sid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = sid / nblk, blk = sid - bh * nblk;
    const int b = bh / H, h = bh % H;
    const int64_t rs = (int64_t)3 * H * HD;
    const bf16_t* qb = qkv + (int64_t)b * L * rs + (int64_t)h * HD;
    const bf16_t* kb = qb + (int64_t)H * HD;
    const bf16_t* vb = kb + (int64_t)H * HD;
    // queries q_begin .. L-1 only (q_begin > 0: the last block of a stack, whose other output rows nobody reads); their
    // outputs go to a COMPACT o [B, L - q_begin, H, HD]; keys are always all L rows; lse2 keeps the full [B, H, L] index
    const int q0 = q_begin + blk * 128 + wave * 32;
    const int Lq = L - q_begin;

    constexpr int TILE = AG<HD>::TILE, KS = AG<HD>::KS, DT = AG<HD>::DT;
    bf16x8 qf[KS];
    load_own<KS>(qb, rs, q0, L, lane, qf);

    f32x16 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m = -__builtin_inff(), lsum = 0.f;

    const int nt = (L + 63) / 64;
    // LDS: [buffer 0: K | V][buffer 1: K | V]
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const TileAddr<HD> ad = tile_addr<HD>(lane, sbase);
    stage64<HD>(kb, rs, 0, L, sbase, tid, wave);
    stage64<HD>(vb, rs, 0, L, sbase + TILE, tid, wave);
    pin_loaded(qf);
    dma_drain();
    __syncthreads();

    // hot loop: full 64-key tiles only; a ragged last tile (L % 64 != 0) runs once, after the loop, so its masking
    // code never shares registers with the steady state
    const int nfull = (L & 63) ? nt - 1 : nt;
    unsigned soff[AG<HD>::CH / 4];
    stage_offsets<HD>(rs, tid, soff);
    if constexpr (CAUSAL) {
        const int q_end = min(L, q_begin + blk * 128 + 128);            // one past the workgroup's last query
        const int nt_c = min(nt, (q_end + 63) / 64);                    // tiles any of its queries can see
        const int nvis = min(nfull, (q0 + 1) / 64);                     // tiles with every key <= the wave's first query
        const int lim = min(L, q0 + (lane & 31) + 1);                   // this lane's query sees keys < lim
        for (int t = 0; t < nt_c; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt_c) {
                if (t + 1 < nfull) {
                    const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
                    stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
                    stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
                } else {
                    stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
                    stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
                }
            }
            const unsigned kl = cur * 2 * TILE;
            if (t < nvis) fwd_tile<HD, false>(kl, kl + TILE, ad, qf, oacc, m, lsum, t * 64, L, scale_log2e, half);
            else if (t * 64 <= q0 + 31) fwd_tile<HD, true>(kl, kl + TILE, ad, qf, oacc, m, lsum, t * 64, lim, scale_log2e, half);
            dma_drain();
            __syncthreads();
        }
    } else {
    // two tiles per trip so that the LDS buffer of a tile body is a compile-time constant (an instruction immediate, not an add)
    auto stage_next = [&](int t, int cur) {   // tile t + 1 into the other buffer
        if (t + 1 < nfull) {          // next tile is a full one: scalar base + invariant lane offsets
            const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
            stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
            stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
        } else if (t + 1 < nt) {      // the ragged last tile: clamped rows
            stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
            stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
        }
    };
    VT_STAMP_DECL;
    for (int t = 0; t < nfull; t += 2) {
        VT_STAMP_START;
        stage_next(t, 0);
        fwd_tile<HD, false>(0, TILE, ad, qf, oacc, m, lsum, t * 64, L, scale_log2e, half);
        VT_STAMP(0);
        dma_drain();
        VT_STAMP(1);
        __syncthreads();
        VT_STAMP(2);
        VT_STAMP_ITER;
        if (t + 1 < nfull) {
            stage_next(t + 1, 1);
            fwd_tile<HD, false>(2 * TILE, 3 * TILE, ad, qf, oacc, m, lsum, (t + 1) * 64, L, scale_log2e, half);
            VT_STAMP(0);
            dma_drain();
            VT_STAMP(1);
            __syncthreads();
            VT_STAMP(2);
            VT_STAMP_ITER;
        }
    }
    VT_STAMP_FLUSH(0);
    if (nfull < nt) {
        const unsigned kl = (nfull & 1) * 2 * TILE;
        fwd_tile<HD, true>(kl, kl + TILE, ad, qf, oacc, m, lsum, nfull * 64, L, scale_log2e, half);
    }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
    const int q = q0 + (lane & 31);
    const bool ok = q < L;
    store_own<DT>(oacc, 1.0f / ltot, o + (int64_t)b * Lq * H * HD + (int64_t)h * HD, (int64_t)H * HD, q - q_begin, ok, half);
    if (ok && half == 0) lse2[((int64_t)b * H + h) * L + q] = m + __builtin_amdgcn_logf(ltot);  // v_log_f32 = log2
}

template <int HD, bool TAIL>
__device__ __forceinline__ void dq_tile(unsigned kl, unsigned vl, const TileAddr<HD>& ad, const bf16x8 (&qf)[HD / 16], const bf16x8 (&dof)[HD / 16],
                                        f32x16 (&dq)[HD / 32], float my_lse, float my_delta, int key0, int L, float c, int half) {
    const f32x2 c2 = {c, c}, nl2 = {-my_lse, -my_lse}, nd2 = {-my_delta, -my_delta};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag_a<HD>(kl, ad, kt * 32, s), qf[s], sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag_a<HD>(vl, ad, kt * 32, s), dof[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; r += 2) {   // two elements per instruction: v_pk_fma_f32, v_pk_add_f32, v_pk_mul_f32
            const f32x2 t = __builtin_elementwise_fma((f32x2){sacc[r], sacc[r + 1]}, c2, nl2);
            f32x2 p = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
            if (TAIL) {
                if (key0 + kt * 32 + reg_row(r, half) >= L) p[0] = 0.f;
                if (key0 + kt * 32 + reg_row(r + 1, half) >= L) p[1] = 0.f;
            }
            const f32x2 ds = p * ((f32x2){dp[r], dp[r + 1]} + nd2);  // dS (unscaled)
            sacc[r] = ds[0];
            sacc[r + 1] = ds[1];
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 dsf = pack8(sacc, sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_a<HD>(kl, ad, kt * 32, sp, dt), dsf, dq[dt], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: own rows = queries; streams K (row reads + transposed reads) and V (row reads).  Also produces
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d] for its own rows (both operands are one 16-B load per k-step away) and
// leaves it in `delta` for the dK/dV kernel, which is launched behind this one.
// ------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL = false>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                              const bf16_t* __restrict__ dO, const float* __restrict__ lse2,
                                                              float* __restrict__ delta, bf16_t* __restrict__ dqkv, int L, int H, int nblk,
                                                              float scale, float scale_log2e, int q_begin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int sid = xcd_remap(blockIdx.x, gridDim.x);  // see attn_fwd_kernel
    const int bh = sid / nblk, blk = sid - bh * nblk;
    const int b = bh / H, h = bh % H;
    const int64_t rs = (int64_t)3 * H * HD, ors = (int64_t)H * HD;
    const bf16_t* qb = qkv + (int64_t)b * L * rs + (int64_t)h * HD;
    const bf16_t* kb = qb + (int64_t)H * HD;
    const bf16_t* vb = kb + (int64_t)H * HD;
    const int q0 = q_begin + blk * 128 + wave * 32;   // kept queries only; o / dO are compact [B, L - q_begin, H, HD]
    const int Lq = L - q_begin;
    const int q = q0 + (lane & 31);
    const int qc = q < L ? q : L - 1;

    constexpr int TILE = AG<HD>::TILE, KS = AG<HD>::KS, DT = AG<HD>::DT;
    bf16x8 qf[KS], dof[KS];
    load_own<KS>(qb, rs, q0, L, lane, qf);
    load_own<KS>(dO + (int64_t)b * Lq * ors + (int64_t)h * HD, ors, q0 - q_begin, Lq, lane, dof);
    const float my_lse = lse2[((int64_t)b * H + h) * L + qc];
    float my_delta = 0.f;
    {
        bf16x8 of[KS];
        load_own<KS>(o + (int64_t)b * Lq * ors + (int64_t)h * HD, ors, q0 - q_begin, Lq, lane, of);
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) my_delta += bf2f(of[s][j]) * bf2f(dof[s][j]);
        my_delta += __shfl_xor(my_delta, 32);   // the two lane halves hold the two halves of every 16-wide k-step
        if (q < L && half == 0) delta[((int64_t)b * H + h) * L + q] = my_delta;
    }

    f32x16 dq[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;

    const int nt = (L + 63) / 64;
    // LDS: [buffer 0: K | V][buffer 1: K | V]
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const TileAddr<HD> ad = tile_addr<HD>(lane, sbase);
    stage64<HD>(kb, rs, 0, L, sbase, tid, wave);
    stage64<HD>(vb, rs, 0, L, sbase + TILE, tid, wave);
    pin_loaded(qf);
    pin_loaded(dof);
    float lse_pin = my_lse;
    pin_loaded(lse_pin);
    pin_loaded(my_delta);
    dma_drain();
    __syncthreads();

    const int nfull = (L & 63) ? nt - 1 : nt;
    unsigned soff[AG<HD>::CH / 4];
    stage_offsets<HD>(rs, tid, soff);
    if constexpr (CAUSAL) {   // see attn_fwd_kernel
        const int q_end = min(L, q_begin + blk * 128 + 128);
        const int nt_c = min(nt, (q_end + 63) / 64);
        const int nvis = min(nfull, (q0 + 1) / 64);
        const int lim = min(L, q0 + (lane & 31) + 1);
        for (int t = 0; t < nt_c; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt_c) {
                if (t + 1 < nfull) {
                    const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
                    stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
                    stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
                } else {
                    stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
                    stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
                }
            }
            const unsigned kl = cur * 2 * TILE;
            if (t < nvis) dq_tile<HD, false>(kl, kl + TILE, ad, qf, dof, dq, lse_pin, my_delta, t * 64, L, scale_log2e, half);
            else if (t * 64 <= q0 + 31) dq_tile<HD, true>(kl, kl + TILE, ad, qf, dof, dq, lse_pin, my_delta, t * 64, lim, scale_log2e, half);
            dma_drain();
            __syncthreads();
        }
    } else {
    auto stage_next = [&](int t, int cur) {   // tile t + 1 into the other buffer
        if (t + 1 < nfull) {
            const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
            stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
            stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
        } else if (t + 1 < nt) {
            stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
            stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
        }
    };
    VT_STAMP_DECL;
    for (int t = 0; t < nfull; t += 2) {      // two tiles per trip: the LDS buffer of a body is an instruction immediate
        VT_STAMP_START;
        stage_next(t, 0);
        dq_tile<HD, false>(0, TILE, ad, qf, dof, dq, lse_pin, my_delta, t * 64, L, scale_log2e, half);
        VT_STAMP(0);
        dma_drain();
        VT_STAMP(1);
        __syncthreads();
        VT_STAMP(2);
        VT_STAMP_ITER;
        if (t + 1 < nfull) {
            stage_next(t + 1, 1);
            dq_tile<HD, false>(2 * TILE, 3 * TILE, ad, qf, dof, dq, lse_pin, my_delta, (t + 1) * 64, L, scale_log2e, half);
            VT_STAMP(0);
            dma_drain();
            VT_STAMP(1);
            __syncthreads();
            VT_STAMP(2);
            VT_STAMP_ITER;
        }
    }
    VT_STAMP_FLUSH(1);
    if (nfull < nt) {
        const unsigned kl = (nfull & 1) * 2 * TILE;
        dq_tile<HD, true>(kl, kl + TILE, ad, qf, dof, dq, lse_pin, my_delta, nfull * 64, L, scale_log2e, half);
    }
    }
    store_own<DT>(dq, scale, dqkv + (int64_t)b * L * rs + (int64_t)h * HD, rs, q, q < L, half);
}

// one 64-query tile of the dK/dV sweep.  LDS buffer: Q tile | dO tile | lse2[64] | delta[64]
template <int HD, bool TAIL, bool CAUSAL = false>
__device__ __forceinline__ void dkv_tile(unsigned qt_l, unsigned lse_a, const TileAddr<HD>& ad, const bf16x8 (&kf)[HD / 16], const bf16x8 (&vf)[HD / 16],
                                         f32x16 (&dk)[HD / 32], f32x16 (&dv)[HD / 32], int q0, int L, float c, int half, int my_key = 0) {
    constexpr int TILE = AG<HD>::TILE;
    const unsigned do_l = qt_l + TILE;
    const unsigned lse_l = lse_a + qt_l + 2 * TILE;      // lse_a = LDS base + 16 * half: this lane's 16 query rows are 4 runs of 4 consecutive rows, qt*32 + 8g + 4*half + 0..3
    const f32x2 nc2 = {-c, -c};
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag_a<HD>(qt_l, ad, qt * 32, s), kf[s], sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag_a<HD>(do_l, ad, qt * 32, s), vf[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 lse4 = lds_ld128f(lse_l + (qt * 32 + 8 * g) * 4);
            const f32x4 del4 = lds_ld128f(lse_l + 256 + (qt * 32 + 8 * g) * 4);
#pragma unroll
            for (int e = 0; e < 4; e += 2) {   // two elements per instruction
                const int r = 4 * g + e;
                // p = 2^(s c - lse) as 2^-(lse - s c): the row constant enters the packed fma as it comes out of the LDS (negating it
                // first cost a v_xor per element), the sign rides on v_exp_f32's source modifier
                const f32x2 t = __builtin_elementwise_fma((f32x2){sacc[r], sacc[r + 1]}, nc2, (f32x2){lse4[e], lse4[e + 1]});
                f32x2 p = {__builtin_amdgcn_exp2f(-t[0]), __builtin_amdgcn_exp2f(-t[1])};
                if (TAIL) {
                    if (q0 + qt * 32 + reg_row(r, half) >= L) p[0] = 0.f;
                    if (q0 + qt * 32 + reg_row(r + 1, half) >= L) p[1] = 0.f;
                }
                if (CAUSAL) {     // a query never attends a later key
                    if (q0 + qt * 32 + reg_row(r, half) < my_key) p[0] = 0.f;
                    if (q0 + qt * 32 + reg_row(r + 1, half) < my_key) p[1] = 0.f;
                }
                const f32x2 ds = p * ((f32x2){dp[r], dp[r + 1]} - (f32x2){del4[e], del4[e + 1]});
                sacc[r] = p[0];
                sacc[r + 1] = p[1];
                dp[r] = ds[0];
                dp[r + 1] = ds[1];
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc, sp);
            const bf16x8 dsf = pack8(dp, sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_a<HD>(do_l, ad, qt * 32, sp, dt), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag_a<HD>(qt_l, ad, qt * 32, sp, dt), dsf, dk[dt], 0, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: own rows = keys; streams Q and dO tiles (both row reads and transposed reads) + lse2/delta
// ------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL = false>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 3) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                               const float* __restrict__ lse2, const float* __restrict__ delta,
                                                               bf16_t* __restrict__ dqkv, int L, int H, int nblk, float scale, float scale_log2e,
                                                               int q_begin) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int sid = xcd_remap(blockIdx.x, gridDim.x);  // key blocks of one head share the Q/dO stream: keep them on one XCD
    const int bh = sid / nblk, blk = sid - bh * nblk;
    const int b = bh / H, h = bh % H;
    const int64_t rs = (int64_t)3 * H * HD, ors = (int64_t)H * HD;
    const bf16_t* qb = qkv + (int64_t)b * L * rs + (int64_t)h * HD;
    const bf16_t* kb = qb + (int64_t)H * HD;
    const bf16_t* vb = kb + (int64_t)H * HD;
    // query tiles from q_begin on (a multiple of 64; the rows before it got no gradient), dO compact [B, L - q_begin, H, HD]
    const int Lq = L - q_begin;
    const bf16_t* dob = dO + (int64_t)b * Lq * ors + (int64_t)h * HD;
    const float* lse_b = lse2 + ((int64_t)b * H + h) * L;
    const float* del_b = delta + ((int64_t)b * H + h) * L;
    const int k0 = blk * 128 + wave * 32;
    const int key = k0 + (lane & 31);

    constexpr int TILE = AG<HD>::TILE, KS = AG<HD>::KS, DT = AG<HD>::DT;
    bf16x8 kf[KS], vf[KS];
    load_own<KS>(kb, rs, k0, L, lane, kf);
    load_own<KS>(vb, rs, k0, L, lane, vf);

    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;

    // LDS: per buffer  Q tile | dO tile | lse2[64] | delta[64]
    constexpr int BUF = 2 * TILE + 512;
    const int nt = (L + 63) / 64;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const TileAddr<HD> ad = tile_addr<HD>(lane, sbase);
    unsigned lse_a = sbase + 16 * half;
    asm volatile("" : "+v"(lse_a));
    unsigned qoff[AG<HD>::CH / 4], dooff[AG<HD>::CH / 4];
    stage_offsets<HD>(rs, tid, qoff);
    stage_offsets<HD>(ors, tid, dooff);
    const int nfull_q = L / 64;        // query tiles without a ragged row
    auto stage = [&](int t, int buf) {
        const unsigned base = sbase + buf * BUF;
        if (t < nfull_q) {
            stage64_full<HD>(qb + (int64_t)t * 64 * rs, qoff, base, wave);
            stage64_full<HD>(dob + ((int64_t)t * 64 - q_begin) * ors, dooff, base + TILE, wave);
        } else {
            stage64<HD>(qb, rs, t * 64, L, base, tid, wave);
            stage64<HD>(dob, ors, t * 64 - q_begin, Lq, base + TILE, tid, wave);
        }
        if (wave < 2) {  // wave 0: lse2[64], wave 1: delta[64] by 4-byte LDS-DMA (rows past L clamped; masked at use)
            int qq = t * 64 + lane;
            qq = qq < L ? qq : L - 1;
            glds4_asm((wave == 0 ? lse_b : del_b) + qq, base + 2 * TILE + wave * 256);
        }
    };
    // CAUSAL: query tiles before the workgroup's first key contribute nothing (every query < every key)
    const int t0 = CAUSAL ? max(q_begin >> 6, (blk * 128) >> 6) : q_begin >> 6;
    stage(t0, 0);
    pin_loaded(kf);
    pin_loaded(vf);
    dma_drain();
    __syncthreads();

    const int nfull = (L & 63) ? nt - 1 : nt;
    if constexpr (CAUSAL) {
        for (int t = t0; t < nt; ++t) {
            const int cur = (t - t0) & 1;
            if (t + 1 < nt) stage(t + 1, cur ^ 1);
            if (t < nfull && t * 64 >= k0 + 31) dkv_tile<HD, false>(cur * BUF, lse_a, ad, kf, vf, dk, dv, t * 64, L, scale_log2e, half);
            else if (t * 64 + 63 >= k0) dkv_tile<HD, true, true>(cur * BUF, lse_a, ad, kf, vf, dk, dv, t * 64, L, scale_log2e, half, key);
            dma_drain();
            __syncthreads();
        }
    } else {
    VT_STAMP_DECL;
    for (int t = t0; t < nfull; t += 2) {     // two tiles per trip: the LDS buffer of a body is an instruction immediate
        VT_STAMP_START;
        if (t + 1 < nt) stage(t + 1, 1);
        dkv_tile<HD, false>(0, lse_a, ad, kf, vf, dk, dv, t * 64, L, scale_log2e, half);
        VT_STAMP(0);
        dma_drain();
        VT_STAMP(1);
        __syncthreads();
        VT_STAMP(2);
        VT_STAMP_ITER;
        if (t + 1 < nfull) {
            if (t + 2 < nt) stage(t + 2, 0);
            dkv_tile<HD, false>(BUF, lse_a, ad, kf, vf, dk, dv, (t + 1) * 64, L, scale_log2e, half);
            VT_STAMP(0);
            dma_drain();
            VT_STAMP(1);
            __syncthreads();
            VT_STAMP(2);
            VT_STAMP_ITER;
        }
    }
    VT_STAMP_FLUSH(2);
    if (nfull < nt) dkv_tile<HD, true>(((nfull - t0) & 1) * BUF, lse_a, ad, kf, vf, dk, dv, nfull * 64, L, scale_log2e, half);
    }
    bf16_t* dkb = dqkv + (int64_t)b * L * rs + (int64_t)h * HD + (int64_t)H * HD;
    store_own<DT>(dk, scale, dkb, rs, key, key < L, half);
    store_own<DT>(dv, 1.0f, dkb + (int64_t)H * HD, rs, key, key < L, half);
}

// dQ rows of the queries before q_begin: they received no gradient (the dQ kernel only visits the kept queries)
__global__ void zero_q_rows_kernel(bf16_t* __restrict__ dqkv, int L, int q_begin, int64_t rs, int qcols) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over B * q_begin * qcols / 8
    const int per_row = qcols >> 3;
    const int64_t r = idx / per_row;
    const int c = (int)(idx % per_row) * 8;
    const int64_t b = r / q_begin, q = r % q_begin;
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = f2bf(0.f);
    *(bf16x8*)(dqkv + (b * L + q) * rs + c) = z;
}

}  // namespace

template <int HD>
static void launch_fwd(const void* qkv, int B, int L, int H, int q_begin, void* o, float* lse2, hipStream_t s) {
    const float sl2 = (HD == 64 ? 0.125f : 0.17677669529663688110f) * 1.44269504088896340736f;
    const int nblk = (L - q_begin + 127) / 128;
    hipLaunchKernelGGL(attn_fwd_kernel<HD>, dim3(nblk * B * H), dim3(256), 4 * AG<HD>::TILE, s, (const bf16_t*)qkv, (bf16_t*)o, lse2, L, H, nblk, sl2, q_begin);
}

template <int HD>
static void launch_bwd(const void* qkv, const void* o, const void* dO, const float* lse2, int B, int L, int H, int q_begin, void* dqkv,
                       float* delta_ws, hipStream_t s) {
    const float scale = HD == 64 ? 0.125f : 0.17677669529663688110f, sl2 = scale * 1.44269504088896340736f;
    if (q_begin > 0) {
        const int64_t n = (int64_t)B * q_begin * (H * HD / 8);
        hipLaunchKernelGGL(zero_q_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)dqkv, L, q_begin, (int64_t)3 * H * HD, H * HD);
    }
    const int nblk_q = (L - q_begin + 127) / 128, nblk_k = (L + 127) / 128;
    hipLaunchKernelGGL(attn_bwd_dq_kernel<HD>, dim3(nblk_q * B * H), dim3(256), 4 * AG<HD>::TILE, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dO, lse2,
                       delta_ws, (bf16_t*)dqkv, L, H, nblk_q, scale, sl2, q_begin);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<HD>, dim3(nblk_k * B * H), dim3(256), 2 * (2 * AG<HD>::TILE + 512), s, (const bf16_t*)qkv, (const bf16_t*)dO, lse2,
                       delta_ws, (bf16_t*)dqkv, L, H, nblk_k, scale, sl2, q_begin);
}

// every operand is read and every output written 16 bytes per lane (the outputs since the widened stores of round 5)
static bool attn_aligned(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

static int attn_check(const char* who, int B, int L, int H, int hd, int q_begin) {
    VT_CHECK_ARG(hd == 64 || hd == 32, "%s: head_dim %d unsupported (64 or 32)", who, hd);
    VT_CHECK_ARG(B > 0 && L > 0 && H > 0, "%s: bad shape", who);
    VT_CHECK_ARG(q_begin >= 0 && q_begin < L && q_begin % 64 == 0, "%s: q_begin=%d must be a multiple of 64 below L=%d", who, q_begin, L);
    return VT_OK;
}

extern "C" int vt_attention_fwd_rows(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t q_begin, void* o_compact, float* lse2,
                                     vtStream stream) {
    VT_CHECK_ARG(qkv && o_compact && lse2, "vt_attention_fwd: null pointer");
    VT_CHECK_ARG(attn_aligned(qkv, o_compact), "vt_attention_fwd: qkv and o must be 16-byte aligned");
    int rc = attn_check("vt_attention_fwd", B, L, H, hd, q_begin);
    if (rc) return rc;
    if (hd == 64) launch_fwd<64>(qkv, B, L, H, q_begin, o_compact, lse2, (hipStream_t)stream);
    else launch_fwd<32>(qkv, B, L, H, q_begin, o_compact, lse2, (hipStream_t)stream);
    VT_CHECK_LAUNCH("vt_attention_fwd");
    return VT_OK;
}

extern "C" int vt_attention_bwd_rows(const void* qkv, const void* o_compact, const void* dO_compact, const float* lse2, int32_t B, int32_t L,
                                     int32_t H, int32_t hd, int32_t q_begin, void* dqkv, float* delta_ws, vtStream stream) {
    VT_CHECK_ARG(qkv && o_compact && dO_compact && lse2 && dqkv && delta_ws, "vt_attention_bwd: null pointer");
    VT_CHECK_ARG(attn_aligned(qkv, o_compact, dO_compact, dqkv), "vt_attention_bwd: qkv, o, dO and dqkv must be 16-byte aligned");
    int rc = attn_check("vt_attention_bwd", B, L, H, hd, q_begin);
    if (rc) return rc;
    if (hd == 64) launch_bwd<64>(qkv, o_compact, dO_compact, lse2, B, L, H, q_begin, dqkv, delta_ws, (hipStream_t)stream);
    else launch_bwd<32>(qkv, o_compact, dO_compact, lse2, B, L, H, q_begin, dqkv, delta_ws, (hipStream_t)stream);
    VT_CHECK_LAUNCH("vt_attention_bwd");
    return VT_OK;
}

extern "C" int vt_attention_fwd(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, void* o, float* lse2, vtStream stream) {
    return vt_attention_fwd_rows(qkv, B, L, H, hd, 0, o, lse2, stream);
}

extern "C" int vt_attention_bwd(const void* qkv, const void* o, const void* dO, const float* lse2, int32_t B, int32_t L, int32_t H,
                                int32_t hd, void* dqkv, float* delta_ws, vtStream stream) {
    return vt_attention_bwd_rows(qkv, o, dO, lse2, B, L, H, hd, 0, dqkv, delta_ws, stream);
}


// ------------------------------------------------------------------------------------------------
// causal attention (the AR consumer): F.scaled_dot_product_attention(q, k, v, is_causal=True) of
// /root/reference/models/larp_ar.py:186-190 and its autograd; head_dim 64 (every llama-abs size: dim / n_head = 64)
// ------------------------------------------------------------------------------------------------
extern "C" int vt_attention_causal_fwd(const void* qkv, int32_t B, int32_t L, int32_t H, void* o, float* lse2, vtStream stream) {
    VT_CHECK_ARG(qkv && o && lse2, "vt_attention_causal_fwd: null pointer");
    VT_CHECK_ARG(B > 0 && L > 0 && H > 0, "vt_attention_causal_fwd: bad shape");
    VT_CHECK_ARG(attn_aligned(qkv, o), "vt_attention_causal_fwd: qkv and o must be 16-byte aligned");
    const float sl2 = 0.125f * 1.44269504088896340736f;
    const int nblk = (L + 127) / 128;
    hipLaunchKernelGGL((attn_fwd_kernel<64, true>), dim3(nblk * B * H), dim3(256), 4 * AG<64>::TILE, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)o, lse2, L, H,
                       nblk, sl2, 0);
    VT_CHECK_LAUNCH("vt_attention_causal_fwd");
    return VT_OK;
}

extern "C" int vt_attention_causal_bwd(const void* qkv, const void* o, const void* dO, const float* lse2, int32_t B, int32_t L, int32_t H, void* dqkv,
                                       float* delta_ws, vtStream stream) {
    VT_CHECK_ARG(qkv && o && dO && lse2 && dqkv && delta_ws, "vt_attention_causal_bwd: null pointer");
    VT_CHECK_ARG(B > 0 && L > 0 && H > 0, "vt_attention_causal_bwd: bad shape");
    VT_CHECK_ARG(attn_aligned(qkv, o, dO, dqkv), "vt_attention_causal_bwd: qkv, o, dO and dqkv must be 16-byte aligned");
    const float scale = 0.125f, sl2 = scale * 1.44269504088896340736f;
    const int nblk = (L + 127) / 128;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((attn_bwd_dq_kernel<64, true>), dim3(nblk * B * H), dim3(256), 4 * AG<64>::TILE, s, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dO,
                       lse2, delta_ws, (bf16_t*)dqkv, L, H, nblk, scale, sl2, 0);
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, true>), dim3(nblk * B * H), dim3(256), 2 * (2 * AG<64>::TILE + 512), s, (const bf16_t*)qkv, (const bf16_t*)dO, lse2,
                       delta_ws, (bf16_t*)dqkv, L, H, nblk, scale, sl2, 0);
    VT_CHECK_LAUNCH("vt_attention_causal_bwd");
    return VT_OK;
}

#ifdef VT_ATTN_STAMPS
extern "C" int vt_attention_stamps(unsigned long long* host_out) {   // diagnostic build only: 3 x 2048 x 4 x 6 counters
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), sizeof(g_attn_stamps)) == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
#endif
