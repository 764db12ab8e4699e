// Attention backward in FIVE matrix products (head_dim 64, no mask): dQ, dK and dV of timm Attention's
// F.scaled_dot_product_attention (constructed at /root/reference/models/transformer.py:52-59) from one sweep.
//
// The two-kernel backward of vt_attention.hip recomputes S and dP in both kernels (7 products, two exp passes, every
// K/V and Q/dO tile staged twice).  Here ONE workgroup owns 256 keys of a (batch, head):
//     S  = Q.K^T,  dP = dO.V^T            key on the MFMA lane (own K / V rows are B operands held in registers)
//     P  = exp2(c S - lse2),  dS = P (dP - delta)                                         once
//     dV^T += dO^T.P,  dK^T += Q^T.dS     the accumulators P / dS ARE the B operands (accumulator-as-operand k order)
//     dQ^T  = K^T.dS^T                    sums over the key = the lane index of dS, so dS crosses LDS once: each lane
//                                         stores 4 consecutive queries of its key (8 bytes) into a [key][query] image with
//                                         the tile swizzle, both operands come back by transposed reads
// dK / dV never leave the registers until the end.  dQ of a (batch, head) is the sum over its ceil(L / 256) key blocks:
// an ORDERED HAND-OFF (no atomics, bit-reproducible): for every 64-query slice the key blocks form a chain, each adds its
// tile to its predecessor's sum in a fixed order and the last one writes bf16 dQ.  Slice order is rotated per key block
// (block j starts at slice j * nsl / nkb), the chain of slice s starts at the block whose turn comes first, so in steady
// state a predecessor finished a slice ~nsl / nkb slices before its successor needs it and nobody waits.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, recipe R1 in its counter form):
//   producer: 16-byte write-through (sc1) stores of the partial sum, every storing wave drains vmcnt(0), then adds 1 to the
//             slice's arrival counter (relaxed, agent scope);
//   consumer: polls the counter one slice AHEAD (the value is consumed an iteration later, so its latency is hidden), one
//             agent-scope acquire, then plain LDS-DMA loads of the partial sum into LDS;
//   every spin is bounded by the realtime clock and reports through a status word; counters, the work-queue head and the
//   status word are zeroed by a memset node ahead of every launch.
// Residency: work items (batch-head major, key block minor) are pulled from a queue, so a workgroup holding item k
// implies every item < k is held by a running workgroup: at most the team at the queue's frontier can be incomplete and
// every earlier team finishes without it -- no grid size or placement assumption, no deadlock.
//
// One iteration (a 64-query slice), 8 waves x 32 keys, two barriers:
//   phase A  stage Q/dO of the next slice (LDS-DMA) | S, dP, softmax, dS -> LDS image, dV, dK      (all waves)
//            waves 4-7: predecessor's partial sum of THIS slice by LDS-DMA -> pbuf (acquire first)
//   barrier
//   phase B  dQ^T tile (32 d x 32 q) per wave over its 128-key half; waves 4-7 start from pbuf, write to xbuf
//   barrier
//            waves 0-3 add xbuf to theirs and publish (or write bf16 dQ if last in the chain)
#include "vt_common.h"
#include <stdlib.h>

#include "vt_attn_tile.h"

namespace {

constexpr int FB_T = 8192;                       // bytes of a [64][64] bf16 tile
constexpr int FB_KIMG = 0;                       // K image of the 256 own keys: 4 tiles
constexpr int FB_VIMG = 4 * FB_T;                // V image of the same keys (B operands of dP are read from here, not held in registers)
constexpr int FB_QBUF = 8 * FB_T;                // 2 x (Q tile | dO tile | lse2[64] | delta[64])
constexpr int FB_QSZ = 2 * FB_T + 512;
constexpr int FB_DS = FB_QBUF + 2 * FB_QSZ;      // dS^T image [256 keys][64 queries]
constexpr int FB_XBUF = FB_DS + 4 * FB_T;        // waves 4-7 -> waves 0-3: 4 tiles x 4 KB fp32
constexpr int FB_MISC = FB_XBUF + 16384;         // work-queue ticket broadcast
constexpr int FB_LDS = FB_MISC + 16;

struct FusedBwdArgs {
    const bf16_t* qkv;
    const bf16_t* dO;
    const float* lse2;
    const float* delta;
    bf16_t* dqkv;
    float* P;              // partial dQ sums, [B * H][nsl][4 tiles][4][64 lanes][4] fp32 (fragment order: 1 KB per wave store)
    unsigned* ctr;         // [B * H][nsl] arrival counters
    unsigned* ticket;      // work-queue head
    unsigned* status;      // != 0: a bounded spin gave up (results invalid)
    int L, H, q_begin, nkb, nsl, nitems;
    float scale, scale_log2e;
};

// Diagnostic build only (-DFB_STAMPS, tools/fb_stamps.sh): s_memtime stamps around the segments of an iteration, summed per wave
// into a buffer of their own.  In the real kernel no stamp executes; the diagnostic build's run time is never quoted.
#ifdef FB_STAMPS
__device__ unsigned long long fb_dbg[2048 * 8];
#define FB_STAMP(k)                                                                   \
    do {                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                            \
        unsigned long long t__;                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");   \
        __builtin_amdgcn_sched_barrier(0);                                            \
        fb_acc[k] += t__ - fb_last;                                                   \
        fb_last = t__;                                                                \
    } while (0)
#else
#define FB_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ unsigned long long realtime() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

// 16-byte write-through store (sc1): leaves the XCD's L2 for the memory side, so a consumer on any XCD reads it after its acquire
__device__ __forceinline__ void st_sc1(float* p, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// workgroup barrier that leaves this wave's LDS-DMA / stores in flight (a __syncthreads() would drain vmcnt too)
__device__ __forceinline__ void wg_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 512 threads stage one [64][64] bf16 tile: one 16-byte LDS-DMA per thread, swizzle on the source chunk
__device__ __forceinline__ unsigned stage_off512(int64_t rs, int tid) {
    const int row = tid >> 3;
    const int lc = (tid & 7) ^ fsw<64>(row);
    return (unsigned)((row * rs + lc * 8) * 2);
}
__device__ __forceinline__ void stage512_full(const bf16_t* tile_row0, unsigned off, unsigned lds, int wave) {
    glds16_sv(tile_row0, off, lds + wave * 1024);
}
__device__ __forceinline__ void stage512_clamped(const bf16_t* src, int64_t rs, int row0, int nrows, unsigned lds, int tid, int wave) {
    const int row = tid >> 3;
    const int lc = (tid & 7) ^ fsw<64>(row);
    int gr = row0 + row;
    gr = gr < nrows ? gr : nrows - 1;
    glds16_asm(src + (int64_t)gr * rs + lc * 8, lds + wave * 1024);
}

// phase A of one slice for one wave: S, dP, P, dS (-> LDS image), dV, dK.   TQ: mask queries >= L; TK: mask keys >= L
template <bool TQ, bool TK>
__device__ __forceinline__ void fb_phase_a(const char* qt_l, char* ds_l, const char* k_row, f32x16 (&dk)[2], f32x16 (&dv)[2],
                                           int q0, int L, bool kvalid, int keyrow, float c, int lane, int half) {
    const char* do_l = qt_l + FB_T;
    const float* lse_l = (const float*)(qt_l + 2 * FB_T);
    const float* del_l = lse_l + 64;
    char* ds_row = ds_l + keyrow * 128 + 8 * half;
    const int fs = fsw<64>(keyrow);
    // own K / V rows as the B operands of S = Q.K^T and dP = dO.V^T, k-step s: B[k = 16s + 8*half + j][key] = chunk 2s + half of
    // the image's row (held in LDS, not in registers: the kernel sits at the 256-VGPR limit of two waves per SIMD)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x16 sacc, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[r] = dp[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 kfs = *(const bf16x8*)(k_row + (((2 * s + half) ^ fs) << 4));
            const bf16x8 vfs = *(const bf16x8*)(k_row + (FB_VIMG - FB_KIMG) + (((2 * s + half) ^ fs) << 4));
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<64>(qt_l, qt * 32, s, lane), kfs, sacc, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag<64>(do_l, qt * 32, s, lane), vfs, dp, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 lse4 = *(const f32x4*)(lse_l + qt * 32 + 8 * g + 4 * half);
            const f32x4 del4 = *(const f32x4*)(del_l + qt * 32 + 8 * g + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * g + e;
                float p = __builtin_amdgcn_exp2f(fmaf(sacc[r], c, -lse4[e]));
                if (TQ && (q0 + qt * 32 + reg_row(r, half) >= L)) p = 0.f;
                if (TK && !kvalid) p = 0.f;
                sacc[r] = p;
                dp[r] = p * (dp[r] - del4[e]);
            }
        }
#pragma unroll
        for (int sp = 0; sp < 2; ++sp) {
            const bf16x8 pf = pack8(sacc, sp);
            const bf16x8 dsf = pack8(dp, sp);
            // dS^T image: this lane's key row, queries qt*32 + 8g + 4*half + 0..3 for g = 2sp, 2sp+1 (8 bytes each)
            *(bf16x4*)(ds_row + (((4 * qt + 2 * sp) ^ fs) << 4)) = __builtin_shufflevector(dsf, dsf, 0, 1, 2, 3);
            *(bf16x4*)(ds_row + (((4 * qt + 2 * sp + 1) ^ fs) << 4)) = __builtin_shufflevector(dsf, dsf, 4, 5, 6, 7);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<64>(do_l, qt * 32, sp, dt * 32, lane), pf, dv[dt], 0, 0, 0);
                dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<64>(qt_l, qt * 32, sp, dt * 32, lane), dsf, dk[dt], 0, 0, 0);
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(const FusedBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int L = a.L, H = a.H, nkb = a.nkb, nsl = a.nsl;
    const int64_t rs = (int64_t)3 * H * 64, ors = (int64_t)H * 64;
    const int Lq = L - a.q_begin;
    const unsigned sbase = __builtin_amdgcn_readfirstlane(lds_addr_of(smem));
    const unsigned qoff = stage_off512(rs, tid), dooff = stage_off512(ors, tid);
    const int dqt = wave & 1, qqt = (wave >> 1) & 1, kh = wave >> 2;     // phase B: d tile, query tile, key half of this wave
    const int xoff = ((dqt + 2 * qqt) * 4) * 1024 + lane * 16;            // this wave's tile in pbuf / xbuf / P (fragment order)
    const unsigned long long t_start = realtime();
#ifdef FB_STAMPS
    unsigned long long fb_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fb_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(fb_last)::"memory");
#endif
    constexpr unsigned long long SPIN_LIMIT = 30000000ull;                // 0.3 s of 100 MHz ticks: bounded spins

    for (;;) {
        // ---- pull a work item ----
        __syncthreads();
        if (tid == 0) *(unsigned*)(smem + FB_MISC) = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int item = __builtin_amdgcn_readfirstlane(*(const unsigned*)(smem + FB_MISC));
        if (item >= a.nitems) break;
        const int bh = item / nkb, kb = item - bh * nkb;
        const int b = bh / H, h = bh - b * H;
        const bf16_t* qb = a.qkv + (int64_t)b * L * rs + (int64_t)h * 64;
        const bf16_t* kbp = qb + (int64_t)H * 64;
        const bf16_t* vbp = kbp + (int64_t)H * 64;
        const bf16_t* dob = a.dO + (int64_t)b * Lq * ors + (int64_t)h * 64;
        const float* lse_b = a.lse2 + (int64_t)bh * L;
        const float* del_b = a.delta + (int64_t)bh * L;
        float* Pb = a.P + (int64_t)bh * nsl * 4096;
        unsigned* ctr_b = a.ctr + (int64_t)bh * nsl;
        const int key_base = kb * 256;
        const int k0 = key_base + wave * 32;
        const int key = k0 + (lane & 31);
        const bool kvalid = key < L;
        const bool ragged_k = key_base + 256 > L;          // workgroup-uniform

        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[dt][r] = dv[dt][r] = 0.f;

        // K and V images: 4 tiles of 64 keys each (rows past L clamped: their P / dS columns are zero)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (key_base + 64 * j + 64 <= L) {
                stage512_full(kbp + (int64_t)(key_base + 64 * j) * rs, qoff, sbase + FB_KIMG + j * FB_T, wave);
                stage512_full(vbp + (int64_t)(key_base + 64 * j) * rs, qoff, sbase + FB_VIMG + j * FB_T, wave);
            } else {
                stage512_clamped(kbp, rs, key_base + 64 * j, L, sbase + FB_KIMG + j * FB_T, tid, wave);
                stage512_clamped(vbp, rs, key_base + 64 * j, L, sbase + FB_VIMG + j * FB_T, tid, wave);
            }
        }
        const int s0 = (int)(((unsigned)kb * (unsigned)nsl) / (unsigned)nkb);    // first slice of this key block (rotated start)
        auto stage_slice = [&](int s, int buf) {
            const unsigned base = sbase + FB_QBUF + buf * FB_QSZ;
            const int row0 = a.q_begin + 64 * s;
            if (row0 + 64 <= L) {
                stage512_full(qb + (int64_t)row0 * rs, qoff, base, wave);
                stage512_full(dob + (int64_t)(64 * s) * ors, dooff, base + FB_T, wave);
            } else {
                stage512_clamped(qb, rs, row0, L, base, tid, wave);
                stage512_clamped(dob, ors, 64 * s, Lq, base + FB_T, tid, wave);
            }
            if (wave < 2) {
                int qq = row0 + lane;
                qq = qq < L ? qq : L - 1;
                glds4_asm((wave == 0 ? lse_b : del_b) + qq, base + 2 * FB_T + wave * 256);
            }
        };
        // chain bookkeeping of slice s: head = the key block whose turn at s comes first; position counts down the blocks
        auto chain_pos = [&](int s) {     // head(s) = max{ j : start(j) <= s } = min(nkb - 1, ((s + 1) nkb - 1) / nsl)
            unsigned head = ((unsigned)(s + 1) * (unsigned)nkb - 1u) / (unsigned)nsl;
            head = head < (unsigned)(nkb - 1) ? head : (unsigned)(nkb - 1);
            const int pos = (int)head - kb;
            return pos < 0 ? pos + nkb : pos;
        };
        // bounded wait for the slice's arrival counter (slow path: normally the value polled an iteration ahead already suffices)
        auto wait_counter = [&](const unsigned* c, unsigned want, unsigned have) {
            while (have < want) {
                have = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (have >= want) break;
                if (realtime() - t_start > SPIN_LIMIT) {
                    if (lane == 0) __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
        };
        // predecessor's partial sum of slice s (this wave's tile): four 16-byte sc1 (L1-bypassing) loads to registers.  No acquire:
        // the counter was polled by an sc1 load a slice earlier, workgroup barriers lie between that poll and these loads, every
        // byte was stored sc1 in whole 128-byte lines and drained before its wave's arrival (MI355X_MICROARCH.md, visibility,
        // the third "valid form" row).  An acquire here (buffer_inv sc1) blocks the CU's whole vector-memory path for ~1.7 us per
        // slice -- measured with stamps: 28 % of the kernel.
        const auto p_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Pb, 0, nsl * 16384, 0x00020000);
        f32x4 pv[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) pv[g4] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto fetch_prev = [&](int s) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                pv[g4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(p_rsrc, xoff + g4 * 1024, s * 16384, 16));
        };
        auto next_slice = [&](int s) { return s + 1 >= nsl ? 0 : s + 1; };

        int pos_c = chain_pos(s0);                          // this block's place in the chain of the current slice
        int pos_n = nsl > 1 ? chain_pos(next_slice(s0)) : 0;
        stage_slice(s0, 0);
        unsigned seen = 0;
        if (kh == 1) {
            if (pos_c > 0) {                                // only when several key blocks share a first slice (nkb > nsl)
                wait_counter(ctr_b + s0, 4u * (unsigned)pos_c, 0u);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // poll and loads not separated by a barrier: keep the acquire
                fetch_prev(s0);
            }
            if (nsl > 1 && pos_n > 0) seen = __hip_atomic_load(ctr_b + next_slice(s0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        dma_drain();
        __syncthreads();

        int pending = -1;                                   // slice whose write-through stores are out but not yet announced
        bool have_prev = true;                              // this slice's predecessor sum is (on its way) in pbuf[cur]
        int s = s0;
        for (int i = 0; i < nsl; ++i) {
            const int sn = next_slice(s), sn2 = next_slice(sn);
            const int cur = i & 1;
            const int pos = pos_c;
            const bool last = pos == nkb - 1;
            const int pos_nn = i + 2 < nsl ? chain_pos(sn2) : 0;
            // ---- phase A: next slice's tiles on their way, S / dP / softmax / dS / dV / dK of this one ----
            FB_STAMP(7);
            if (i + 1 < nsl) stage_slice(sn, cur ^ 1);
            {
                int lane_a = lane;
                asm volatile("" : "+v"(lane_a));
                const int half_a = lane_a >> 5;
                const int keyrow_a = wave * 32 + (lane_a & 31);
                const char* qt_l = smem + FB_QBUF + cur * FB_QSZ;
                const char* k_row = smem + FB_KIMG + keyrow_a * 128;
                const int q0 = a.q_begin + 64 * s;
                const bool tq = q0 + 64 > L;
                if (!tq && !ragged_k) fb_phase_a<false, false>(qt_l, smem + FB_DS, k_row, dk, dv, q0, L, kvalid, keyrow_a, a.scale_log2e, lane_a, half_a);
                else if (!tq) fb_phase_a<false, true>(qt_l, smem + FB_DS, k_row, dk, dv, q0, L, kvalid, keyrow_a, a.scale_log2e, lane_a, half_a);
                else fb_phase_a<true, true>(qt_l, smem + FB_DS, k_row, dk, dv, q0, L, kvalid, keyrow_a, a.scale_log2e, lane_a, half_a);
            }
            FB_STAMP(0);
            if (kh == 0 && pending >= 0) {
                // announce the previous slice: its write-through stores were issued a whole phase ago, so this drain does not
                // stall, and it comes BEFORE any wait of this iteration (a block never waits while it owes an announcement:
                // the dependency order of the chains stays the acyclic one of the header comment)
                dma_drain();
                if (lane == 0) __hip_atomic_fetch_add(ctr_b + pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pending = -1;
            }
            FB_STAMP(1);
            wg_barrier();                                   // dS image complete
            FB_STAMP(2);
            // ---- phase B: dQ^T tile of this wave over its 128-key half ----
            // (lane made opaque per phase: hipcc otherwise hoists every fragment address of both phases out of the slice loop
            //  and holds ~40 of them across phase A, which is already at the register limit)
            int lane_b = lane;
            asm volatile("" : "+v"(lane_b));
            f32x16 dq;
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[r] = 0.f;
            if (kh == 1 && pos > 0) {
                if (!have_prev) {                           // slow path: the predecessor was not done when we looked a slice ago
                    wait_counter(ctr_b + s, 4u * (unsigned)pos, 0u);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    fetch_prev(s);
                }
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[4 * g4 + e] = pv[g4][e];
            }
            FB_STAMP(3);
#pragma unroll
            for (int sp = 0; sp < 8; ++sp)
                dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(trfrag<64>(smem + FB_KIMG, 128 * kh, sp, 32 * dqt, lane_b),
                                                             trfrag<64>(smem + FB_DS, 128 * kh, sp, 32 * qqt, lane_b), dq, 0, 0, 0);
            if (kh == 1) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = dq[4 * g4 + e];
                    *(f32x4*)(smem + FB_XBUF + xoff + g4 * 1024) = v;
                }
            }
            dma_drain();                                    // own LDS-DMA of the next slice's tiles landed (kh == 0: and the previous
                                                            // slice's write-through stores are out)
            if (kh == 1) {
                // the NEXT slice's predecessor sum, if its counter (polled a slice ago) already says so: a whole iteration to arrive
                have_prev = false;
                if (i + 1 < nsl && pos_n > 0 && seen >= 4u * (unsigned)pos_n) {
                    fetch_prev(sn);
                    have_prev = true;
                }
                seen = 0;
                if (i + 2 < nsl && pos_nn > 0) seen = __hip_atomic_load(ctr_b + sn2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            FB_STAMP(4);
            wg_barrier();
            FB_STAMP(5);
            if (kh == 0) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = *(const f32x4*)(smem + FB_XBUF + xoff + g4 * 1024);
#pragma unroll
                    for (int e = 0; e < 4; ++e) dq[4 * g4 + e] += v[e];
                }
                if (last) {     // end of the chain: bf16 dQ, scaled
                    const int q = a.q_begin + 64 * s + 32 * qqt + (lane & 31);
                    if (q < L) {
                        bf16_t* p = a.dqkv + ((int64_t)b * L + q) * rs + (int64_t)h * 64 + 32 * dqt + 4 * half;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            bf16x4 v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = f2bf(dq[4 * g4 + e] * a.scale);
                            *(bf16x4*)(p + 8 * g4) = v;
                        }
                    }
                } else {        // write-through stores now, the announcement after the next phase A (their latency stays off the critical path)
                    float* dst = Pb + (int64_t)s * 4096 + (dqt + 2 * qqt) * 1024 + lane * 4;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = dq[4 * g4 + e];
                        st_sc1(dst + g4 * 256, v);
                    }
                    pending = s;
                }
            }
            FB_STAMP(6);
            pos_c = pos_n;
            pos_n = pos_nn;
            s = sn;
        }
        if (kh == 0 && pending >= 0) {
            dma_drain();
            if (lane == 0) __hip_atomic_fetch_add(ctr_b + pending, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- dK (scaled), dV of the own keys ----
        bf16_t* dkb = a.dqkv + (int64_t)b * L * rs + (int64_t)h * 64 + (int64_t)H * 64;
        store_own<2>(dk, a.scale, dkb, rs, key, kvalid, half);
        store_own<2>(dv, 1.0f, dkb + (int64_t)H * 64, rs, key, kvalid, half);
    }
#ifdef FB_STAMPS
    if (lane == 0 && blockIdx.x < 256)
        for (int k = 0; k < 8; ++k) fb_dbg[(blockIdx.x * 8 + wave) * 8 + k] = fb_acc[k];
#endif
}

// delta[b, h, q] = sum_d dO[b, q - q_begin, h, d] * O[...]: 8 lanes x 16 bytes per (row, head)
__global__ void attn_delta_kernel(const bf16_t* __restrict__ o, const bf16_t* __restrict__ dO, float* __restrict__ delta, int64_t n8, int L, int H, int q_begin) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = idx < n8;
    const int64_t i = ok ? idx : n8 - 1;
    const bf16x8 x = *(const bf16x8*)(o + i * 8);
    const bf16x8 y = *(const bf16x8*)(dO + i * 8);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += bf2f(x[j]) * bf2f(y[j]);
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    if (ok && (idx & 7) == 0) {
        const int64_t rh = idx >> 3;              // (b * Lq + qc) * H + h
        const int h = (int)(rh % H);
        const int64_t rq = rh / H;
        const int Lq = L - q_begin;
        const int64_t b = rq / Lq;
        const int qc = (int)(rq - b * Lq);
        delta[(b * H + h) * L + q_begin + qc] = acc;
    }
}

__global__ void zero_q_rows_kernel2(bf16_t* __restrict__ dqkv, int L, int q_begin, int64_t rs, int qcols) {
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

struct FusedPlan {
    int nkb, nsl, nitems;
    size_t p_bytes, ctr_off, ctr_bytes, status_off, total;
};
FusedPlan fused_plan(int B, int L, int H, int q_begin) {
    FusedPlan p;
    p.nkb = (L + 255) / 256;
    p.nsl = (L - q_begin + 63) / 64;
    p.nitems = B * H * p.nkb;
    p.p_bytes = (size_t)B * H * p.nsl * 16384;
    // control block first (zeroed per launch, a multiple of 16 bytes): [ticket, pad, pad, pad][counters]; then the partial sums;
    // the STICKY status word is the last 16 bytes of the caller's workspace (set by a kernel whose bounded spin gave up, never
    // cleared by a launch: the caller zeroes the workspace once, vt_attention_bwd_fused_status / the engine read it at leisure)
    p.ctr_bytes = (((size_t)B * H * p.nsl * 4 + 16) + 15) / 16 * 16;
    p.ctr_off = 0;
    p.status_off = p.ctr_bytes + p.p_bytes;
    p.total = p.status_off + 16;
    return p;
}

}  // namespace

static const int g_cu_count = [] {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
}();

extern "C" size_t vt_attention_bwd_fused_workspace_bytes(int32_t B, int32_t L, int32_t H, int32_t hd, int32_t q_begin) {
    if (B <= 0 || L <= 0 || H <= 0 || hd != 64 || q_begin < 0 || q_begin >= L || q_begin % 64) return 0;
    return fused_plan(B, L, H, q_begin).total;
}

extern "C" int vt_attention_bwd_fused(const void* qkv, const void* o_compact, const void* dO_compact, const float* lse2, int32_t B, int32_t L, int32_t H,
                                      int32_t hd, int32_t q_begin, void* dqkv, float* delta_ws, void* ws, size_t ws_bytes, vtStream stream) {
    VT_CHECK_ARG(qkv && o_compact && dO_compact && lse2 && dqkv && delta_ws && ws, "vt_attention_bwd_fused: null pointer");
    VT_CHECK_ARG(hd == 64, "vt_attention_bwd_fused: head_dim %d unsupported (64)", hd);
    VT_CHECK_ARG(B > 0 && L > 0 && H > 0, "vt_attention_bwd_fused: bad shape");
    VT_CHECK_ARG(q_begin >= 0 && q_begin < L && q_begin % 64 == 0, "vt_attention_bwd_fused: q_begin=%d must be a multiple of 64 below L=%d", q_begin, L);
    const FusedPlan p = fused_plan(B, L, H, q_begin);
    VT_CHECK_ARG(ws_bytes >= p.total, "vt_attention_bwd_fused: workspace %zu < %zu bytes", ws_bytes, p.total);
    VT_CHECK_ARG(((uintptr_t)ws & 15) == 0, "vt_attention_bwd_fused: workspace must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int Lq = L - q_begin;
    if (hipMemsetAsync(ws, 0, p.ctr_bytes, s) != hipSuccess) {
        vt_set_error("vt_attention_bwd_fused: memset failed");
        return VT_ERR_LAUNCH;
    }
    if (q_begin > 0) {
        const int64_t n = (int64_t)B * q_begin * (H * 64 / 8);
        hipLaunchKernelGGL(zero_q_rows_kernel2, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (bf16_t*)dqkv, L, q_begin, (int64_t)3 * H * 64, H * 64);
    }
    const float c_log2 = 0.125f * 1.44269504088896340736f;
    const int64_t n8 = (int64_t)B * Lq * H * 8;
    hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, (const bf16_t*)o_compact, (const bf16_t*)dO_compact, delta_ws, n8, L, H,
                       q_begin);
    FusedBwdArgs a;
    a.qkv = (const bf16_t*)qkv;
    a.dO = (const bf16_t*)dO_compact;
    a.lse2 = lse2;
    a.delta = delta_ws;
    a.dqkv = (bf16_t*)dqkv;
    unsigned* ctl = (unsigned*)ws;
    a.ticket = ctl;
    a.ctr = ctl + 4;
    a.status = (unsigned*)((char*)ws + (ws_bytes / 16) * 16 - 16);   // the LAST 16 bytes of whatever the caller supplied
    a.P = (float*)((char*)ws + p.ctr_bytes);
    a.L = L;
    a.H = H;
    a.q_begin = q_begin;
    a.nkb = p.nkb;
    a.nsl = p.nsl;
    a.nitems = p.nitems;
    a.scale = 0.125f;
    a.scale_log2e = c_log2;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)attn_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FB_LDS) != hipSuccess) {
            vt_set_error("vt_attention_bwd_fused: hipFuncSetAttribute(%d bytes of LDS) failed", FB_LDS);
            return VT_ERR_LAUNCH;
        }
        attr_set = true;
    }
    const int grid = p.nitems < g_cu_count ? p.nitems : g_cu_count;
    hipLaunchKernelGGL(attn_bwd_fused_kernel, dim3(grid), dim3(512), FB_LDS, s, a);
    VT_CHECK_LAUNCH("vt_attention_bwd_fused");
    return VT_OK;
}

// sticky status word of this workspace = its last 16 bytes (0 = no bounded spin ever gave up); synchronises the stream
extern "C" int vt_attention_bwd_fused_status(const void* ws, size_t ws_bytes, int32_t* status, vtStream stream) {
    VT_CHECK_ARG(ws && status && ws_bytes >= 32, "vt_attention_bwd_fused_status: bad argument");
    unsigned v = 0;
    if (hipMemcpyAsync(&v, (const char*)ws + (ws_bytes / 16) * 16 - 16, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        vt_set_error("vt_attention_bwd_fused_status: copy failed");
        return VT_ERR_LAUNCH;
    }
    *status = (int32_t)v;
    return VT_OK;
}

#ifdef FB_STAMPS
extern "C" int vt_attention_bwd_fused_stamps(unsigned long long* host_out) {   // [256 workgroups][8 waves][8 segments] cycles
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(fb_dbg), sizeof(unsigned long long) * 2048 * 8) == hipSuccess ? VT_OK : VT_ERR_LAUNCH;
}
#endif
