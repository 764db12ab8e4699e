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

namespace {

// one 64-key tile of the forward: S^T = K.Q^T, online softmax (log2 domain; max taken on the raw scores since
// the scale is positive), O^T += V^T.P^T.  TAIL masks keys >= L (last tile of a ragged sequence only).
template <int HD, bool TAIL>
__device__ __forceinline__ void fwd_tile(const char* kl, const char* vl, const bf16x8 (&qf)[HD / 16], f32x16 (&oacc)[HD / 32], float& m, float& lsum,
                                         int key0, int L, float c, int lane, int half) {
    f32x16 sacc[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s)
            sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<HD>(kl, kt * 32, s, lane), qf[s], sacc[kt], 0, 0, 0);
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
    const float mn = m;
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], c, -mn));
            sacc[kt][r] = p;
            ps += p;
        }
    lsum += ps;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc[kt], sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
                oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<HD>(vl, kt * 32, sp, dt * 32, lane), pf, oacc[dt], 0, 0, 0);
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
            const char* kl = smem + cur * 2 * TILE;
            if (t < nvis) fwd_tile<HD, false>(kl, kl + TILE, qf, oacc, m, lsum, t * 64, L, scale_log2e, lane, half);
            else if (t * 64 <= q0 + 31) fwd_tile<HD, true>(kl, kl + TILE, qf, oacc, m, lsum, t * 64, lim, scale_log2e, lane, half);
            dma_drain();
            __syncthreads();
        }
    } else {
    for (int t = 0; t < nfull; ++t) {
        const int cur = t & 1;
        if (t + 1 < nfull) {          // next tile is a full one: scalar base + invariant lane offsets
            const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
            stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
            stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
        } else if (t + 1 < nt) {      // the ragged last tile: clamped rows
            stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
            stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
        }
        const char* kl = smem + cur * 2 * TILE;
        fwd_tile<HD, false>(kl, kl + TILE, qf, oacc, m, lsum, t * 64, L, scale_log2e, lane, half);
        dma_drain();
        __syncthreads();
    }
    if (nfull < nt) {
        const char* kl = smem + (nfull & 1) * 2 * TILE;
        fwd_tile<HD, true>(kl, kl + TILE, qf, oacc, m, lsum, nfull * 64, L, scale_log2e, lane, half);
    }
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
    const int q = q0 + (lane & 31);
    const bool ok = q < L;
    store_own<DT>(oacc, 1.0f / ltot, o + (int64_t)b * Lq * H * HD + (int64_t)h * HD, (int64_t)H * HD, q - q_begin, ok, half);
    if (ok && half == 0) lse2[((int64_t)b * H + h) * L + q] = m + __builtin_amdgcn_logf(ltot);  // v_log_f32 = log2
}

// ------------------------------------------------------------------------------------------------
// forward, software-pipelined inside the wave (head_dim 64)
//
// The plain kernel above runs S = K.Q^T, the softmax and O += V^T.P one after the other, so a wave's matrix instructions
// and its vector instructions never overlap (PMC, round 1: matrix pipe 27 % busy + vector issue 55 % of the kernel, the
// SUM of the two was the run time; co-resident waves did not fill the gaps).  Here the unit of work is a 32-key HALF tile
// and every iteration v issues, in program order,
//     4 MFMAs  S(v+1) = K_half(v+1) . Q^T        (scores of the NEXT half)
//     4 MFMAs  O^T   += V_half(v-1)^T . P(v-1)   (accumulate the PREVIOUS half)
// with the 16 exponentials / row sums / bf16 packs of half v -- which depend on neither -- placed between them, two
// elements per MFMA, in source order fenced by sched_barrier(0).  An MFMA keeps the SIMD's vector issue for 8 of its 32
// cycles, so ~6 vector instructions ride in its shadow.
//
// LDS ring: K and V each two 64-key buffers.  Iteration 2t reads K(t) rows 32..63 and V(t-1) rows 32..63; iteration
// 2t+1 reads K(t+1) rows 0..31 and V(t) rows 0..31, and issues the DMA of K(t+2) -> K buffer t&1 and V(t+1) -> V buffer
// (t+1)&1, both last read in iteration 2t: ONE vmcnt(0) + barrier per 64 keys, after the even iteration.
//
// Lazy rescaling with a pipeline: the decision for half v+1 is taken at the end of iteration v (its scores are ready),
// the new reference point is used for the exponentials of v+1, and O / l -- which by then also hold half v, exponentiated
// at the OLD reference -- are multiplied by alpha at the end of iteration v+1 ("scale everything still at the old
// reference exactly once").  A ragged last tile is not pipelined: fwd_tile<TAIL> after the loop.
// ------------------------------------------------------------------------------------------------
// max / sum of a value with its partner lane (lane ^ 32) without an index register or an LDS round trip:
// v_permlane32_swap exchanges lanes 32..63 of the first operand with lanes 0..31 of the second
__device__ __forceinline__ float xhalf_max(float x) {
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

struct FwdState {
    float m;          // reference point of the exponentials (log2 domain), per query = per lane
    float lsum;       // row sum at the scale O is at
    float alpha_p;    // pending multiplier for O / lsum
    bool pend;        // wave-uniform
};

// Per-lane LDS byte offsets of the fragments, relative to a tile image: everything that depends on the buffer, the 32-row
// half or the k-step is a compile-time constant added on top (the swizzle f(row) has period 8 rows), so it lands in the
// instruction's offset field and the loop carries 8 address registers instead of ~50 hoisted address computations.
struct PipeAddr {
    unsigned k[4];       // K row fragment of k-step s: row lane&31
    unsigned v[2][2];    // V transposed fragment, [dt][rows +0 / +8]
};
__device__ __forceinline__ PipeAddr pipe_addr(int lane) {
    PipeAddr a;
    const int row = lane & 31, hf = lane >> 5;
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) a.k[s_] = row * 128 + (((2 * s_ + hf) ^ fsw<64>(row)) << 4);
    const int g = lane >> 4, lam = lane & 15;
    const int r0 = 4 * (g >> 1) + (lam >> 2), r1 = r0 + 8;
    const int bo = (lam & 1) << 3;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const int lc = ((dt * 32 + 16 * (g & 1)) >> 3) + ((lam & 3) >> 1);
        a.v[dt][0] = r0 * 128 + ((lc ^ fsw<64>(r0)) << 4) + bo;
        a.v[dt][1] = r1 * 128 + ((lc ^ fsw<64>(r1)) << 4) + bo;
    }
    return a;
}
constexpr int PIPE_TILE = 64 * 128;   // bytes of a [64][64] bf16 tile; LDS = [K0][V0][K1][V1]
// `pa.k` point into the K buffer in use, `pa.v` into the V buffer in use (the buffer base is part of the register: the
// kernel flips it with one XOR per address and tile); R0 = first row of the 32-row half, a compile-time constant
template <int R0>
__device__ __forceinline__ bf16x8 pipe_kfrag(const char* smem, const PipeAddr& pa, int s_) {
    return *(const bf16x8*)(smem + pa.k[s_] + R0 * 128);
}
template <int R0>
__device__ __forceinline__ bf16x8 pipe_vfrag(const char* smem, const PipeAddr& pa, int i) {   // i = 2 * sp + dt
    const int sp = i >> 1, dt = i & 1;
    const int cst = PIPE_TILE + (R0 + 16 * sp) * 128;
    return cat4(lds_read_tr16(smem + pa.v[dt][0] + cst), lds_read_tr16(smem + pa.v[dt][1] + cst));
}

// one iteration: S_next = K rows [KR, KR+32) . Q^T (if QK), O += V rows [VR, VR+32) . p_prev (if PV), and the softmax of
// s_cur -> p_cur (bf16 B fragments of the two 16-key k-steps), row sum into lsum
// DBG (timing ablations, WRONG results; tools/attn_ablate.sh): 1 no exponentials, 2 no S MFMAs, 4 no PV MFMAs, 8 no LDS fragment
// reads, 16 no softmax arithmetic at all
template <bool QK, bool PV, int KR, int VR, int DBG = 0>
__device__ __forceinline__ void fwd_pipe_iter(const char* smem, const PipeAddr& ad, const bf16x8 (&qf)[4], f32x16& s_next, f32x16& s_cur,
                                              bf16x8 (&p)[2], f32x16 (&oacc)[2], FwdState& st, float c) {
    // p: on entry the packed P of the previous half (B operands of the PV MFMAs), on exit those of this half.  Each k-step
    // is re-packed right behind the last MFMA that reads the old one, so P never needs a second register set.
    // Fragments are read two MFMAs ahead of their use into rotating registers.  hipcc is free to move pure arithmetic and
    // MFMAs anywhere (it SINKS the softmax below the rescale branch and clusters the MFMAs when left alone), so the
    // interleave is pinned by empty asm statements: PIN_F(frag) in front of an MFMA makes the MFMA wait for that point,
    // PIN_P after a pair of exponentials keeps them above it; asm volatile statements keep their relative order.
    bf16x8 fa, fb, fc, fd;
#define VT_PIN_F(f) asm volatile("" : "+v"(f))
    if (QK) { fa = ((DBG & 8) ? qf[0] : pipe_kfrag<KR>(smem, ad, 0)); fb = ((DBG & 8) ? qf[0] : pipe_kfrag<KR>(smem, ad, 1)); }
    else if (PV) { fc = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 0)); fd = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 1)); }
    // The kernel is bound by vector ISSUE (PMC: SQ_ACTIVE_INST_VALU 74 % of the kernel, one quad-cycle per plain VALU
    // instruction and two per v_exp_f32, whatever the number of resident waves), so the arithmetic around the exponentials
    // is done two elements per instruction: v_pk_fma_f32 for s * c - m, v_pk_add_f32 for the row sum.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 nm2 = {-st.m, -st.m}, c2 = {c, c};
    f32x2 ps2 = {0.f, 0.f};
#define VT_SM2(r)                                                            \
    if (!(DBG & 16)) {                                                       \
        const f32x2 sv = {s_cur[r], s_cur[(r) + 1]};                         \
        const f32x2 t = __builtin_elementwise_fma(sv, c2, nm2);              \
        f32x2 pv = {(DBG & 1) ? t[0] : __builtin_amdgcn_exp2f(t[0]), (DBG & 1) ? t[1] : __builtin_amdgcn_exp2f(t[1])}; \
        ps2 += pv;                                                           \
        asm volatile("" : "+v"(pv), "+v"(ps2));                              \
        s_cur[r] = pv[0];                                                    \
        s_cur[(r) + 1] = pv[1];                                              \
        __builtin_amdgcn_sched_barrier(0);                                   \
    }
    if (QK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s_next[r] = 0.f;
        VT_PIN_F(fa);
        if (!(DBG & 2)) s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, qf[0], s_next, 0, 0, 0);
        fa = ((DBG & 8) ? qf[0] : pipe_kfrag<KR>(smem, ad, 2));
    }
    VT_SM2(0);
    if (QK) {
        VT_PIN_F(fb);
        if (!(DBG & 2)) s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, qf[1], s_next, 0, 0, 0);
        fb = ((DBG & 8) ? qf[0] : pipe_kfrag<KR>(smem, ad, 3));
    }
    VT_SM2(2);
    if (QK) {
        VT_PIN_F(fa);
        if (!(DBG & 2)) s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, qf[2], s_next, 0, 0, 0);
        if (PV) fc = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 0));
    }
    VT_SM2(4);
    if (QK) {
        VT_PIN_F(fb);
        if (!(DBG & 2)) s_next = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, qf[3], s_next, 0, 0, 0);
        if (PV) fd = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 1));
    }
    VT_SM2(6);
    if (PV) {
        VT_PIN_F(fc);
        if (!(DBG & 4)) oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc, p[0], oacc[0], 0, 0, 0);
        fc = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 2));
    }
    VT_SM2(8);
    if (PV) {
        VT_PIN_F(fd);
        if (!(DBG & 4)) oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd, p[0], oacc[1], 0, 0, 0);
        fd = ((DBG & 8) ? qf[1] : pipe_vfrag<VR>(smem, ad, 3));
    }
    p[0] = pack8(s_cur, 0);
    VT_SM2(10);
    if (PV) {
        VT_PIN_F(fc);
        if (!(DBG & 4)) oacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fc, p[1], oacc[0], 0, 0, 0);
    }
    VT_SM2(12);
    if (PV) {
        VT_PIN_F(fd);
        if (!(DBG & 4)) oacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fd, p[1], oacc[1], 0, 0, 0);
    }
    VT_SM2(14);
    p[1] = pack8(s_cur, 1);
#undef VT_SM2
#undef VT_PIN_F
    // O and lsum now hold half v-1 too: bring them to the reference the exponentials above used
    if (st.pend) {
        const float a = st.alpha_p;
        st.lsum *= a;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[dt][r] *= a;
        st.pend = false;
    }
    st.lsum += ps2[0] + ps2[1];
    if (QK) {   // decision for the half whose scores just arrived
        float mx = fmaxf(s_next[0], s_next[1]);
#pragma unroll
        for (int r = 2; r < 16; ++r) mx = fmaxf(mx, s_next[r]);
        mx = xhalf_max(mx);
        const float want = mx * c;
        if (__builtin_amdgcn_ballot_w64(want > st.m + 8.0f) != 0ull) {
            const float mn = fmaxf(st.m, want);
            st.alpha_p = __builtin_amdgcn_exp2f(st.m - mn);
            st.m = mn;
            st.pend = true;
        }
    }
    __builtin_amdgcn_sched_barrier(0);   // the next iteration's reads / MFMAs stay behind this one (register pressure)
}

// NW waves per workgroup (4 or 8), 32 queries each: 8 waves share one K/V stream, which halves the L2 -> LDS traffic per query
template <int DBG, int NW>
__global__ __launch_bounds__(64 * NW, 4) void attn_fwd_pipe_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse2,
                                                                int L, int H, int nblk, float scale_log2e, int q_begin) {
    constexpr int HD = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int sid = xcd_remap(blockIdx.x, gridDim.x);
    const int bh = sid / nblk, blk = sid - bh * nblk;
    const int b = bh / H, h = bh % H;
    const int64_t rs = (int64_t)3 * H * HD;
    const bf16_t* qb = qkv + (int64_t)b * L * rs + (int64_t)h * HD;
    const bf16_t* kb = qb + (int64_t)H * HD;
    const bf16_t* vb = kb + (int64_t)H * HD;
    const int q0 = q_begin + blk * (32 * NW) + wave * 32;
    const int Lq = L - q_begin;

    constexpr int TILE = AG<HD>::TILE;
    static_assert(TILE == PIPE_TILE, "tile image size");
    bf16x8 qf[4];
    load_own<4>(qb, rs, q0, L, lane, qf);
    f32x16 oacc[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;

    const int nt = (L + 63) / 64;
    const int nfull = L / 64;                 // tiles the pipeline handles; a ragged last tile (nt == nfull + 1) runs after it
    // LDS: [K buffer 0][V buffer 0][K buffer 1][V buffer 1]
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    constexpr int NTH = 64 * NW, PIECES = 512 / NTH;   // 16-B DMA pieces per thread and tile
    unsigned soff[PIECES];
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int slot = i * NTH + tid, row = slot >> 3;
        soff[i] = (unsigned)((row * rs + (((slot & 7) ^ fsw<HD>(row)) << 3)) * 2);
    }
    auto stage_full = [&](const bf16_t* tile_row0, unsigned lds) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) glds16_sv(tile_row0, soff[i], lds + (i * NTH + wave * 64) * 16);
    };
    auto stage_clamped = [&](const bf16_t* src, int row0, unsigned lds) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int slot = i * NTH + tid, row = slot >> 3;
            int gr = row0 + row;
            gr = gr < L ? gr : L - 1;
            glds16_asm(src + (int64_t)gr * rs + (((slot & 7) ^ fsw<HD>(row)) << 3), lds + (i * NTH + wave * 64) * 16);
        }
    };
    // full tiles only inside the pipeline (scalar base + invariant lane offsets); the ragged last tile, if any, is staged
    // with clamped rows after the loop -- its 64-bit per-lane address arithmetic would otherwise live in the hot loop
    auto stage_k = [&](int t) { stage_full(kb + (int64_t)t * 64 * rs, sbase + (t & 1) * 2 * TILE); };
    auto stage_v = [&](int t) { stage_full(vb + (int64_t)t * 64 * rs, sbase + (t & 1) * 2 * TILE + TILE); };
    if (nfull > 0) {
        stage_k(0);
        stage_v(0);
        if (nfull > 1) stage_k(1);
    } else {
        stage_clamped(kb, 0, sbase);
        stage_clamped(vb, 0, sbase + TILE);
    }
    pin_loaded(qf);
    dma_drain();
    __syncthreads();

    FwdState st;
    st.m = -__builtin_inff();
    st.lsum = 0.f;
    st.alpha_p = 1.f;
    st.pend = false;
    const float c = scale_log2e;
    if (nfull > 0) {
        PipeAddr ad = pipe_addr(lane);      // K addresses -> K buffer 0, V addresses -> V buffer 0
        f32x16 sa, sb;                      // scores of the current / next half (they swap roles every iteration)
        bf16x8 pk[2];
        {   // S(0) and its reference point
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pipe_kfrag<0>(smem, ad, s_), qf[s_], sa, 0, 0, 0);
            float mx = fmaxf(sa[0], sa[1]);
#pragma unroll
            for (int r = 2; r < 16; ++r) mx = fmaxf(mx, sa[r]);
            mx = xhalf_max(mx);
            st.m = mx * c;
        }
        // iteration 0 (tile 0, second half of K): nothing to accumulate yet
        fwd_pipe_iter<true, false, 32, 32, DBG>(smem, ad, qf, sb, sa, pk, oacc, st, c);
        dma_drain();
        __syncthreads();
        // one trip per tile t: odd iteration 2t+1 (scores sb, previous P pa; K(t+1) rows 0.., V(t) rows 0..), then even
        // iteration 2t+2 (scores sa, previous P pb; K(t+1) rows 32.., V(t) rows 32..).  K addresses point at buffer
        // (t+1)&1, V addresses at buffer t&1: both flip at the end of the trip.
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) ad.k[s_] ^= 2 * PIPE_TILE;
        for (int t = 0; t + 1 < nfull; ++t) {
            if (t + 2 < nfull) stage_k(t + 2);
            stage_v(t + 1);
            fwd_pipe_iter<true, true, 0, 0, DBG>(smem, ad, qf, sa, sb, pk, oacc, st, c);
            fwd_pipe_iter<true, true, 32, 32, DBG>(smem, ad, qf, sb, sa, pk, oacc, st, c);
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) ad.k[s_] ^= 2 * PIPE_TILE;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                ad.v[dt][0] ^= 2 * PIPE_TILE;
                ad.v[dt][1] ^= 2 * PIPE_TILE;
            }
            dma_drain();
            __syncthreads();
        }
        {   // last full tile (t = nfull - 1): no further scores; accumulate both of its halves and leave the pipeline
            if (nfull < nt) {   // the ragged tile: both of its buffers were last read before the loop's final barrier
                stage_clamped(kb, nfull * 64, sbase + (nfull & 1) * 2 * TILE);
                stage_clamped(vb, nfull * 64, sbase + (nfull & 1) * 2 * TILE + TILE);
            }
            fwd_pipe_iter<false, true, 0, 0, DBG>(smem, ad, qf, sa, sb, pk, oacc, st, c);
#pragma unroll
            for (int i = 0; i < 4; ++i)
                oacc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pipe_vfrag<32>(smem, ad, i), pk[i >> 1], oacc[i & 1], 0, 0, 0);
        }
    }
    float m = st.m, lsum = st.lsum;
    // Everything the epilogue needs is recomputed from an opaque copy of the thread index: values kept live across the tile
    // loop only for these few stores cost registers the loop has no room for (they were spilled to scratch otherwise).
    int tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, half2 = lane2 >> 5;
    if (nfull < nt) {
        if (nfull > 0) {        // K(nfull) / V(nfull) were issued inside the loop (or the prologue): make them visible
            dma_drain();
            __syncthreads();
        }
        const char* kl = smem + (nfull & 1) * 2 * TILE;
        fwd_tile<HD, true>(kl, kl + TILE, qf, oacc, m, lsum, nfull * 64, L, scale_log2e, lane2, half2);
    }
    const float ltot = lsum + __shfl_xor(lsum, 32);
    const int q = q_begin + blk * (32 * NW) + (tid2 >> 6) * 32 + (lane2 & 31);
    const bool ok = q < L;
    store_own<2>(oacc, 1.0f / ltot, o + (int64_t)b * Lq * H * HD + (int64_t)h * HD, (int64_t)H * HD, q - q_begin, ok, half2);
    if (ok && half2 == 0) lse2[((int64_t)b * H + h) * L + q] = m + __builtin_amdgcn_logf(ltot);
}

template <int HD, bool TAIL>
__device__ __forceinline__ void dq_tile(const char* kl, const char* vl, const bf16x8 (&qf)[HD / 16], const bf16x8 (&dof)[HD / 16], f32x16 (&dq)[HD / 32],
                                        float my_lse, float my_delta, int key0, int L, float c, int lane, int half) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<HD>(kl, kt * 32, s, lane), qf[s], sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<HD>(vl, kt * 32, s, lane), dof[s], dp, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -my_lse));
            if (TAIL && (key0 + kt * 32 + reg_row(r, half) >= L)) p = 0.f;
            sacc[r] = p * (dp[r] - my_delta);  // dS (unscaled)
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 dsf = pack8(sacc, sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt)
                dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<HD>(kl, kt * 32, sp, dt * 32, lane), dsf, dq[dt], 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dQ: own rows = queries; streams K (row reads + transposed reads) and V (row reads).  Also produces
// delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d] for its own rows (both operands are one 16-B load per k-step away) and
// leaves it in `delta` for the dK/dV kernel, which is launched behind this one.
// ------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL = false>
__global__ __launch_bounds__(256, CAUSAL ? 2 : 3) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
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
            const char* kl = smem + cur * 2 * TILE;
            if (t < nvis) dq_tile<HD, false>(kl, kl + TILE, qf, dof, dq, lse_pin, my_delta, t * 64, L, scale_log2e, lane, half);
            else if (t * 64 <= q0 + 31) dq_tile<HD, true>(kl, kl + TILE, qf, dof, dq, lse_pin, my_delta, t * 64, lim, scale_log2e, lane, half);
            dma_drain();
            __syncthreads();
        }
    } else {
    for (int t = 0; t < nfull; ++t) {
        const int cur = t & 1;
        if (t + 1 < nfull) {
            const bf16_t* kt = kb + (int64_t)(t + 1) * 64 * rs;
            stage64_full<HD>(kt, soff, sbase + (cur ^ 1) * 2 * TILE, wave);
            stage64_full<HD>(kt + (int64_t)H * HD, soff, sbase + (cur ^ 1) * 2 * TILE + TILE, wave);
        } else if (t + 1 < nt) {
            stage64<HD>(kb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE, tid, wave);
            stage64<HD>(vb, rs, (t + 1) * 64, L, sbase + (cur ^ 1) * 2 * TILE + TILE, tid, wave);
        }
        const char* kl = smem + cur * 2 * TILE;
        dq_tile<HD, false>(kl, kl + TILE, qf, dof, dq, lse_pin, my_delta, t * 64, L, scale_log2e, lane, half);
        dma_drain();
        __syncthreads();
    }
    if (nfull < nt) {
        const char* kl = smem + (nfull & 1) * 2 * TILE;
        dq_tile<HD, true>(kl, kl + TILE, qf, dof, dq, lse_pin, my_delta, nfull * 64, L, scale_log2e, lane, half);
    }
    }
    store_own<DT>(dq, scale, dqkv + (int64_t)b * L * rs + (int64_t)h * HD, rs, q, q < L, half);
}

// one 64-query tile of the dK/dV sweep.  LDS buffer: Q tile | dO tile | lse2[64] | delta[64]
template <int HD, bool TAIL, bool CAUSAL = false>
__device__ __forceinline__ void dkv_tile(const char* qt_l, const bf16x8 (&kf)[HD / 16], const bf16x8 (&vf)[HD / 16], f32x16 (&dk)[HD / 32],
                                         f32x16 (&dv)[HD / 32], int q0, int L, float c, int lane, int half, int my_key = 0) {
    constexpr int TILE = AG<HD>::TILE;
    const char* do_l = qt_l + TILE;
    const float* lse_l = (const float*)(qt_l + 2 * TILE);
    const float* del_l = lse_l + 64;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < HD / 16; ++s) {
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<HD>(qt_l, qt * 32, s, lane), kf[s], sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<HD>(do_l, qt * 32, s, lane), vf[s], dp, 0, 0, 0);
        }
        // this lane's 16 query rows are 4 runs of 4 consecutive rows: rows qt*32 + 8g + 4*half + 0..3
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 lse4 = *(const f32x4*)(lse_l + qt * 32 + 8 * g + 4 * half);
            const f32x4 del4 = *(const f32x4*)(del_l + qt * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -lse4[e]));
                if (TAIL && (q0 + qt * 32 + reg_row(r, half) >= L)) p = 0.f;
                if (CAUSAL && (q0 + qt * 32 + reg_row(r, half) < my_key)) p = 0.f;     // a query never attends a later key
                sacc[r] = p;
                dp[r] = p * (dp[r] - del4[e]);
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc, sp);
            const bf16x8 dsf = pack8(dp, sp);
#pragma unroll
            for (int dt = 0; dt < HD / 32; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<HD>(do_l, qt * 32, sp, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<HD>(qt_l, qt * 32, sp, dt * 32, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// dK, dV: own rows = keys; streams Q and dO tiles (both row reads and transposed reads) + lse2/delta
// ------------------------------------------------------------------------------------------------
template <int HD, bool CAUSAL = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
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
            if (t < nfull && t * 64 >= k0 + 31) dkv_tile<HD, false>(smem + cur * BUF, kf, vf, dk, dv, t * 64, L, scale_log2e, lane, half);
            else if (t * 64 + 63 >= k0) dkv_tile<HD, true, true>(smem + cur * BUF, kf, vf, dk, dv, t * 64, L, scale_log2e, lane, half, key);
            dma_drain();
            __syncthreads();
        }
    } else {
    for (int t = t0; t < nfull; ++t) {
        const int cur = (t - t0) & 1;
        if (t + 1 < nt) stage(t + 1, cur ^ 1);
        dkv_tile<HD, false>(smem + cur * BUF, kf, vf, dk, dv, t * 64, L, scale_log2e, lane, half);
        dma_drain();
        __syncthreads();
    }
    if (nfull < nt) dkv_tile<HD, true>(smem + ((nfull - t0) & 1) * BUF, kf, vf, dk, dv, nfull * 64, L, scale_log2e, lane, half);
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

// Experiment switch for tools/ (environment, read once).  VT_ATTN_PIPE=1 selects the software-pipelined forward below instead of
// the plain one.  Measured (tools/ab_attn.sh, tools/attn_ablate.sh, profiles/r02_attention_*): the pipelined kernel issues its MFMAs
// and exponentials perfectly interleaved and runs within +-3 % of the plain kernel on every box (84.6 vs 81.7 us, 86.6 vs 89.5 us),
// 8 waves per workgroup (half the K/V staging traffic) is 10 % slower, so the plain kernel stays the default.
static const bool g_attn_plain_fwd = [] { const char* e = getenv("VT_ATTN_PIPE"); return !(e && e[0] == '1'); }();
static const int g_attn_waves = [] { const char* e = getenv("VT_ATTN_WAVES"); return e ? atoi(e) : 4; }();   // waves per workgroup of the pipelined forward
static const int g_attn_dbg = [] { const char* e = getenv("VT_ATTN_DBG"); return e ? atoi(e) : 0; }();   // timing ablations (wrong results)

template <int HD>
static void launch_fwd(const void* qkv, int B, int L, int H, int q_begin, void* o, float* lse2, hipStream_t s) {
    const float sl2 = (HD == 64 ? 0.125f : 0.17677669529663688110f) * 1.44269504088896340736f;
    const int nblk = (L - q_begin + 127) / 128;
    if constexpr (HD == 64) {
        if (!g_attn_plain_fwd) {
            const int nw = g_attn_waves == 4 ? 4 : 8;
            const int nb = (L - q_begin + 32 * nw - 1) / (32 * nw);
#define VT_LAUNCH_PIPE(D, W) hipLaunchKernelGGL((attn_fwd_pipe_kernel<D, W>), dim3(nb * B * H), dim3(64 * W), 4 * AG<HD>::TILE, s, (const bf16_t*)qkv, (bf16_t*)o, lse2, L, H, nb, sl2, q_begin)
#define VT_LAUNCH_PIPE_W(D) if (nw == 4) VT_LAUNCH_PIPE(D, 4); else VT_LAUNCH_PIPE(D, 8)
            switch (g_attn_dbg) {
                case 1: VT_LAUNCH_PIPE_W(1); break;
                case 4: VT_LAUNCH_PIPE_W(4); break;
                case 16: VT_LAUNCH_PIPE_W(16); break;
                case 30: VT_LAUNCH_PIPE_W(30); break;
                default: VT_LAUNCH_PIPE_W(0); break;
            }
#undef VT_LAUNCH_PIPE_W
#undef VT_LAUNCH_PIPE
            return;
        }
    }
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

static int attn_check(const char* who, int B, int L, int H, int hd, int q_begin) {
    VT_CHECK_ARG(hd == 64 || hd == 32, "%s: head_dim %d unsupported (64 or 32)", who, hd);
    VT_CHECK_ARG(B > 0 && L > 0 && H > 0, "%s: bad shape", who);
    VT_CHECK_ARG(q_begin >= 0 && q_begin < L && q_begin % 64 == 0, "%s: q_begin=%d must be a multiple of 64 below L=%d", who, q_begin, L);
    return VT_OK;
}

extern "C" int vt_attention_fwd_rows(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t q_begin, void* o_compact, float* lse2,
                                     vtStream stream) {
    VT_CHECK_ARG(qkv && o_compact && lse2, "vt_attention_fwd: null pointer");
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
