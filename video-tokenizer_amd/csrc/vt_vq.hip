// Vector-quantisation codebook search for gfx950 (replaces SimpleVectorQuantizer.forward,
// /root/reference/models/bottleneck.py:262-324, and what autograd derives for its backward).
//
// The reference materialises an N x K fp32 score matrix (268 MB at N = K = 8192) two to four times;
// here the scores live only in MFMA accumulators.  Exact fp32: v_mfma_f32_32x32x2_f32 computes a
// k-ordered chain of fp32 FMAs, so the d-long dot product is bit-identical to the sequential
// fmaf chain that oracle/vq_oracle.c defines, and the VALU is left free for the running arg-best.
//
//   vq_prep_codebook : e = w / max(|w|, eps)   -> E[K,d], E^T[d][Kp] (coalesced LDS staging), |e|^2, |w|
//   vq_prep_tokens   : z = z_in / max(|z_in|, eps), |z|^2
//   vq_search        : workgroup = 256 tokens (4 waves x 2 tiles of 32) x one slice of the codebook;
//                      codebook slices staged through LDS in 128-code chunks (double-buffered);
//                      D[code][token] tiles of 32x32; every lane keeps (best, index) for its token over
//                      the codes it sees in ascending order; lowest index wins ties (torch.argmin/argmax)
//   vq_finalize      : merge slices (ascending), gather q = E[idx], regularized_z = z + (q - z),
//                      squared-error partials (fixed order -> deterministic losses)
//   vq_backward_*    : dz_in and the dense codebook gradient (per-code scan, no atomics, deterministic)
#include "vt_common.h"

namespace {

constexpr int CHUNK = 128;  // codes per LDS chunk

__device__ __forceinline__ float chain_sq(const float* v, int d) {
    float s = 0.f;
    for (int k = 0; k < d; ++k) s = __fmaf_rn(v[k], v[k], s);
    return s;
}

__global__ void vq_prep_codebook_kernel(const float* __restrict__ w, int K, int Kp, int d, int normalize, float* __restrict__ E,
                                        float* __restrict__ ET, float* __restrict__ ee, float* __restrict__ wnorm) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= Kp) return;
    if (k >= K) {  // padding codes: zeros (masked in the search)
        for (int j = 0; j < d; ++j) ET[(int64_t)j * Kp + k] = 0.f;
        ee[k] = 0.f;
        return;
    }
    const float* wr = w + (int64_t)k * d;
    float den = 1.0f;
    if (normalize) {
        const float n = __fsqrt_rn(chain_sq(wr, d));
        den = n > 1e-12f ? n : 1e-12f;
    }
    float s = 0.f;
    for (int j = 0; j < d; ++j) {
        const float e = normalize ? __fdiv_rn(wr[j], den) : wr[j];
        E[(int64_t)k * d + j] = e;
        ET[(int64_t)j * Kp + k] = e;
        s = __fmaf_rn(e, e, s);
    }
    ee[k] = s;
    wnorm[k] = den;
}

__global__ void vq_prep_tokens_kernel(const float* __restrict__ zin, int64_t ldz, int N, int d, int normalize, float* __restrict__ zn,
                                      float* __restrict__ znorm, float* __restrict__ zz) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* zr = zin + (int64_t)n * ldz;
    float den = 1.0f;
    if (normalize) {
        const float nn = __fsqrt_rn(chain_sq(zr, d));
        den = nn > 1e-12f ? nn : 1e-12f;
    }
    float s = 0.f;
    for (int j = 0; j < d; ++j) {
        const float v = normalize ? __fdiv_rn(zr[j], den) : zr[j];
        zn[(int64_t)n * d + j] = v;
        s = __fmaf_rn(v, v, s);
    }
    zz[n] = s;
    znorm[n] = den;
}

__device__ __forceinline__ unsigned pcg_hash(unsigned v) {
    const unsigned state = v * 747796405u + 2891336453u;
    const unsigned word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}

// Gumbel(0,1) noise from a counter-based hash of (seed, token, code): argmax(logit + G) is a draw from
// softmax(logit) == torch.multinomial(softmax(.), 1)  (bottleneck.py:276-280), in one pass.
// The per-call counter (seed_lo) is hashed on its own before the token index is added: with token ^ seed_lo the noise vector of
// (token t, call s) was that of (t ^ s ^ s', call s'), i.e. every call re-used the same N noise vectors, permuted over tokens.
__device__ __forceinline__ float gumbel(unsigned seed_lo, unsigned seed_hi, unsigned token, unsigned code) {
    const unsigned h = pcg_hash(pcg_hash(pcg_hash(seed_lo) + token) + code * 0x9E3779B1u + seed_hi);
    const float u = ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
    return -__logf(-__logf(u));
}

// MODE 0: L2 argmin  score = fl(fl(|z|^2+|e|^2) - 2 z.e)   (lower is better)
// MODE 1: cosine/temperature argmax  score = (z.e) * inv_tau (higher is better)
// MODE 2: as 1 plus Gumbel noise (stochastic sampling)
template <int D2, int MODE>
__global__ __launch_bounds__(256) void vq_search_kernel(const float* __restrict__ zn, const float* __restrict__ zz,
                                                         const float* __restrict__ ET, const float* __restrict__ ee, int N, int K,
                                                         int Kp, float inv_tau, unsigned seed_lo, unsigned seed_hi,
                                                         int chunks_per_split, float* __restrict__ pscore, int* __restrict__ pidx,
                                                         const unsigned* __restrict__ seed_ctr) {
    constexpr int D = 2 * D2;
    constexpr int UNITS = (D * (CHUNK / 4) + 255) / 256;  // float4 staging units per thread
    __shared__ __attribute__((aligned(16))) float lds_e[2][D][CHUNK];
    __shared__ __attribute__((aligned(16))) float lds_ee[2][CHUNK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tl = lane & 31, h = lane >> 5;
    const int tok_base = blockIdx.x * 256 + wave * 64;
    const int split = blockIdx.y;
    // per-call counter kept on the DEVICE (a replayed hipGraph cannot change a by-value seed): added to the low seed word
    if (MODE == 2 && seed_ctr) seed_lo += *seed_ctr;
    const int chunk_begin = split * chunks_per_split;
    const int nchunks_total = Kp / CHUNK;
    const int chunk_end = chunk_begin + chunks_per_split < nchunks_total ? chunk_begin + chunks_per_split : nchunks_total;

    // B operand (tokens), resident for the whole sweep: bt[tile][s] = z[token][2s + h]
    float bt[2][D2];
    float zzv[2];
    int tokv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int tok = tok_base + t * 32 + tl;
        tokv[t] = tok;
        tok = tok < N ? tok : N - 1;
#pragma unroll
        for (int s = 0; s < D2; ++s) bt[t][s] = zn[(int64_t)tok * D + 2 * s + h];
        zzv[t] = zz[tok];
    }
    float best[2];
    int bidx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        best[t] = (MODE == 0) ? __builtin_inff() : -__builtin_inff();
        bidx[t] = 0x7fffffff;
    }

    f32x4 pre[UNITS];
    f32x4 pre_ee = {0.f, 0.f, 0.f, 0.f};
    auto prefetch = [&](int chunk) {
        const int c0 = chunk * CHUNK;
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int unit = u * 256 + tid;
            if (unit < D * (CHUNK / 4)) pre[u] = *(const f32x4*)(ET + (int64_t)(unit >> 5) * Kp + c0 + (unit & 31) * 4);
        }
        if (tid < CHUNK / 4) pre_ee = *(const f32x4*)(ee + c0 + tid * 4);
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int unit = u * 256 + tid;
            if (unit < D * (CHUNK / 4)) *(f32x4*)(&lds_e[buf][unit >> 5][(unit & 31) * 4]) = pre[u];
        }
        if (tid < CHUNK / 4) *(f32x4*)(&lds_ee[buf][tid * 4]) = pre_ee;
    };

    if (chunk_begin < chunk_end) {
        prefetch(chunk_begin);
        commit(0);
    }
    __syncthreads();

    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const int cur = (chunk - chunk_begin) & 1;
        const bool more = chunk + 1 < chunk_end;
        if (more) prefetch(chunk + 1);
#pragma unroll 1
        for (int ct = 0; ct < CHUNK / 32; ++ct) {
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
            for (int s = 0; s < D2; ++s) {
                const float a = lds_e[cur][2 * s + h][ct * 32 + tl];  // A operand: E[code = tl][k = 2s + h]
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt[0][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt[1][s], acc1, 0, 0, 0);
            }
            const int code_base = chunk * CHUNK + ct * 32 + 4 * h;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2);
                const int code = code_base + cl;
                const bool valid = code < K;
                float s0, s1;
                if constexpr (MODE == 0) {
                    const float e2 = lds_ee[cur][ct * 32 + 4 * h + cl];
                    s0 = __fmaf_rn(-2.0f, acc0[r], zzv[0] + e2);
                    s1 = __fmaf_rn(-2.0f, acc1[r], zzv[1] + e2);
                    if (valid && s0 < best[0]) { best[0] = s0; bidx[0] = code; }
                    if (valid && s1 < best[1]) { best[1] = s1; bidx[1] = code; }
                } else {
                    s0 = acc0[r] * inv_tau;
                    s1 = acc1[r] * inv_tau;
                    if constexpr (MODE == 2) {
                        s0 += gumbel(seed_lo, seed_hi, (unsigned)tokv[0], (unsigned)code);
                        s1 += gumbel(seed_lo, seed_hi, (unsigned)tokv[1], (unsigned)code);
                    }
                    if (valid && s0 > best[0]) { best[0] = s0; bidx[0] = code; }
                    if (valid && s1 > best[1]) { best[1] = s1; bidx[1] = code; }
                }
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
    }

    // merge the two lane halves (same token, interleaved code sets): better score, then lower index
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float os = __shfl_xor(best[t], 32);
        const int oi = __shfl_xor(bidx[t], 32);
        const bool take = (MODE == 0) ? (os < best[t] || (os == best[t] && oi < bidx[t]))
                                      : (os > best[t] || (os == best[t] && oi < bidx[t]));
        if (take) { best[t] = os; bidx[t] = oi; }
        if (h == 0 && tokv[t] < N) {
            pscore[(int64_t)split * N + tokv[t]] = best[t];
            pidx[(int64_t)split * N + tokv[t]] = bidx[t];
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void vq_finalize_kernel(const float* __restrict__ pscore, const int* __restrict__ pidx, int nsplit,
                                                           const float* __restrict__ zn, const float* __restrict__ E, int N, int d,
                                                           int64_t* __restrict__ idx_out, float* __restrict__ rz,
                                                           bf16_t* __restrict__ rz_pad, int64_t ldp, float* __restrict__ partial_sq) {
    __shared__ float red[4];
    const int n = blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;
    if (n < N) {
        float best = pscore[n];
        int bi = pidx[n];
        for (int s = 1; s < nsplit; ++s) {  // slices ascend in code index: strict compare keeps the lowest index
            const float v = pscore[(int64_t)s * N + n];
            const int i = pidx[(int64_t)s * N + n];
            const bool take = (MODE == 0) ? (v < best) : (v > best);
            if (take) { best = v; bi = i; }
        }
        if (bi < 0 || bi == 0x7fffffff) bi = 0;  // every score NaN: keep the gather in bounds
        idx_out[n] = bi;
        const float* e = E + (int64_t)bi * d;
        const float* z = zn + (int64_t)n * d;
        for (int j = 0; j < d; ++j) {
            const float diff = e[j] - z[j];
            const float r = z[j] + diff;  // z + (q - z).detach()   (bottleneck.py:307)
            rz[(int64_t)n * d + j] = r;
            if (rz_pad) rz_pad[(int64_t)n * ldp + j] = f2bf(r);
            sq = __fmaf_rn(diff, diff, sq);
        }
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial_sq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// losses[0..3] = loss_q, loss_commit, loss_codebook, mean squared error
__global__ void vq_loss_kernel(const float* __restrict__ partial_sq, int nblk, float inv_count, float beta, float cbw,
                               float* __restrict__ losses) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nblk; ++i) s += (double)partial_sq[i];
    const float mse = (float)(s * (double)inv_count);
    losses[0] = beta * mse + cbw * mse;
    losses[1] = mse;
    losses[2] = mse;
    losses[3] = mse;
}

__global__ void vq_gather_kernel(const float* __restrict__ E, const int64_t* __restrict__ idx, int N, int K, int d, float* __restrict__ out,
                                 bf16_t* __restrict__ out_pad, int64_t ldp) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int64_t i = idx[n];
    i = i < 0 ? 0 : (i >= K ? K - 1 : i);
    for (int j = 0; j < d; ++j) {
        const float v = E[i * d + j];
        if (out) out[(int64_t)n * d + j] = v;
        if (out_pad) out_pad[(int64_t)n * ldp + j] = f2bf(v);
    }
}

// gscal = {d loss_q, d loss_commit, d loss_codebook} (device).  dz = g + s_c*2(z-q)/M; through F.normalize:
// dz_in = (dz - z (z.dz)) / |z_in|
__global__ void vq_bwd_tokens_kernel(const float* __restrict__ g_rz, int64_t ldg, const float* __restrict__ gscal, float beta, float cbw,
                                     const float* __restrict__ zn, const float* __restrict__ znorm, const float* __restrict__ E,
                                     const int64_t* __restrict__ idx, int N, int d, int normalize, float* __restrict__ dz_in,
                                     bf16_t* __restrict__ dz_pad, int64_t ldp) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float s_c = (gscal ? gscal[0] * beta + gscal[1] : 0.f) * 2.0f / ((float)N * (float)d);
    const float* z = zn + (int64_t)n * d;
    const float* e = E + idx[n] * d;
    float dot = 0.f;
    for (int j = 0; j < d; ++j) {
        const float dz = (g_rz ? g_rz[(int64_t)n * ldg + j] : 0.f) + s_c * (z[j] - e[j]);
        dot += z[j] * dz;
    }
    const float inv = 1.0f / znorm[n];
    for (int j = 0; j < d; ++j) {
        const float dz = (g_rz ? g_rz[(int64_t)n * ldg + j] : 0.f) + s_c * (z[j] - e[j]);
        const float v = normalize ? (dz - z[j] * dot) * inv : dz;
        if (dz_in) dz_in[(int64_t)n * d + j] = v;
        if (dz_pad) dz_pad[(int64_t)n * ldp + j] = f2bf(v);
    }
}

// Dense codebook gradient (F.embedding sparse=False through F.normalize), deterministic, atomic-free and INDEPENDENT OF
// HOW THE TOKENS SPREAD OVER THE CODES: de[k] = sum_n [idx[n] == k] (e_k - z_n) is a product OneHot^T . (Q - Z) and
// runs on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: products with 1.0 / 0.0 are exact, the accumulation is a
// sequential fp32 chain over ascending n).  A membership scan per code (the first version) was 10 us when the tokens
// spread evenly and 390 us when they sat on one code -- which is what early training and random weights look like.
//   grid (ceil(K/128), NS): a wave owns 32 codes x one slab of tokens; 64-token chunks of (q_n - z_n) rows are staged
//   in LDS once per workgroup; a wave skips every token pair none of whose indices falls in its 32 codes, so the
//   evenly-spread case costs ~one MFMA per member.  Slab partials [NS][K][d] are summed in ascending slab order by
//   vq_bwd_codebook_finalize, which also applies the scale and the normalisation Jacobian
//   dW[k] = (de - e (e.de)) / |w_k|.
constexpr int CB_CHUNK = 64;
__global__ __launch_bounds__(256) void vq_bwd_codebook_mfma_kernel(const float* __restrict__ zn, const float* __restrict__ E,
                                                                    const int64_t* __restrict__ idx, int N, int K, int d, int slab_len,
                                                                    float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float vals[CB_CHUNK][32];
    __shared__ int idxs[CB_CHUNK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, col = lane & 31;
    const int k0 = (blockIdx.x * 4 + wave) * 32;
    const int n_begin = blockIdx.y * slab_len;
    const int n_end = min(N, n_begin + slab_len);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int tok = tid >> 2, q8 = (tid & 3) * 8;   // staging role: token of the chunk, 8 of its 32 (padded) dims
    for (int c0 = n_begin; c0 < n_end; c0 += CB_CHUNK) {
        {
            const int n = c0 + tok;
            const int code = n < n_end ? (int)idx[n] : -1;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (code >= 0 && q8 < d) {
                const f32x4* e4 = (const f32x4*)(E + (int64_t)code * d + q8);
                const f32x4* z4 = (const f32x4*)(zn + (int64_t)n * d + q8);
                v0 = e4[0] - z4[0];
                v1 = e4[1] - z4[1];
            }
            *(f32x4*)&vals[tok][q8] = v0;
            *(f32x4*)&vals[tok][q8 + 4] = v1;
            if ((tid & 3) == 0) idxs[tok] = code;
        }
        __syncthreads();
        const unsigned long long mine = __ballot((unsigned)(idxs[lane] - k0) < 32u);   // tokens of this chunk on my 32 codes
        if (mine) {
#pragma unroll 4
            for (int t = 0; t < CB_CHUNK / 2; ++t) {
                if ((mine >> (2 * t)) & 3ull) {
                    const float a = (idxs[2 * t + half] == k0 + col) ? 1.0f : 0.0f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, vals[2 * t + half][col], acc, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    if (col < d) {
        float* out = part + ((int64_t)blockIdx.y * K) * d + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (k < K) out[(int64_t)k * d] = acc[r];
        }
    }
}

// 32 lanes per code: lane j sums the slab partials of dimension j in ascending slab order
__global__ __launch_bounds__(256) void vq_bwd_codebook_finalize(const float* __restrict__ part, int nslab, const float* __restrict__ gscal,
                                                                 float cbw, const float* __restrict__ E, const float* __restrict__ wnorm,
                                                                 int N, int K, int d, int normalize, float* __restrict__ dW) {
#pragma clang fp contract(off)   // every rounding below is spelled out: the product e*de must not fuse into the first butterfly add
    const int k = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int j = threadIdx.x & 31;
    const float s_b = (gscal ? gscal[0] * cbw + gscal[2] : 0.f) * 2.0f / ((float)N * (float)d);
    float de = 0.f, e = 0.f;
    if (k < K && j < d) {
        for (int s = 0; s < nslab; ++s) de += part[((int64_t)s * K + k) * d + j];
        de *= s_b;
        e = E[(int64_t)k * d + j];
    }
    float dot = e * de;
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    // one fused multiply-add, written out so the order is defined: the oracle restates it bit for bit (oracle/vq_oracle.c)
    if (k < K && j < d) dW[(int64_t)k * d + j] = normalize ? fmaf(-e, dot, de) / wnorm[k] : de;
}

static inline int cb_slabs(int N) {
    int ns = (N + 511) / 512;
    return ns < 1 ? 1 : (ns > 32 ? 32 : ns);
}

}  // namespace

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

static void vq_split_plan(int N, int Kp, int* nsplit, int* chunks_per_split) {
    const int tblocks = (N + 255) / 256;
    const int nchunks = Kp / CHUNK;
    int want = 768 / tblocks;
    if (want < 1) want = 1;
    if (want > nchunks) want = nchunks;
    const int cps = (nchunks + want - 1) / want;
    *chunks_per_split = cps;
    *nsplit = (nchunks + cps - 1) / cps;
}

// scratch layout: ET[d*Kp] | ee[Kp] | zz[N] | pscore[S*N] | pidx[S*N] | partial_sq[ceil(N/256)]
extern "C" size_t vt_vq_workspace_bytes(int32_t N, int32_t K, int32_t d) {
    const int Kp = round_up(K, CHUNK);
    int S, cps;
    vq_split_plan(N, Kp, &S, &cps);
    size_t f = (size_t)d * Kp + Kp + N + 2 * (size_t)S * N + (N + 255) / 256 + 64;
    const size_t bwd = (size_t)cb_slabs(N) * K * d;  // slab partials of the codebook gradient (vt_vq_backward with dW != NULL)
    return (f > bwd ? f : bwd) * 4;
}

extern "C" int vt_vq_forward_ctr(const float* z_in, int64_t ldz, const float* codebook, int32_t N, int32_t K, int32_t d, int32_t mode,
                                 int32_t l2_normalized, float inv_tau, float beta, float codebook_w, uint64_t seed, const uint32_t* seed_counter,
                                 float* E, float* wnorm, float* zn, float* znorm, int64_t* idx, float* rz, void* rz_pad_bf16, int64_t ldp,
                                 float* losses, void* workspace, vtStream stream) {
    VT_CHECK_ARG(z_in && codebook && E && wnorm && zn && znorm && idx && rz && losses && workspace, "vt_vq_forward: null pointer");
    VT_CHECK_ARG(N > 0 && K > 0 && (d == 8 || d == 16 || d == 24 || d == 32), "vt_vq_forward: d=%d must be 8,16,24 or 32", d);
    VT_CHECK_ARG(mode >= 0 && mode <= 2, "vt_vq_forward: mode must be 0 (l2 argmin), 1 (cos argmax) or 2 (cos sample)");
    VT_CHECK_ARG(mode == 0 || l2_normalized, "vt_vq_forward: cosine modes require l2 normalisation (bottleneck.py:274)");
    VT_CHECK_ARG(!rz_pad_bf16 || ldp >= d, "vt_vq_forward: ldp < d");
    hipStream_t s = (hipStream_t)stream;
    const int Kp = round_up(K, CHUNK);
    int S, cps;
    vq_split_plan(N, Kp, &S, &cps);
    float* ET = (float*)workspace;
    float* ee = ET + (size_t)d * Kp;
    float* zz = ee + Kp;
    float* pscore = zz + N;
    int* pidx = (int*)(pscore + (size_t)S * N);
    float* partial = (float*)(pidx + (size_t)S * N);
    const int nblk = (N + 255) / 256;

    hipLaunchKernelGGL(vq_prep_codebook_kernel, dim3((Kp + 255) / 256), dim3(256), 0, s, codebook, K, Kp, d, l2_normalized, E, ET, ee, wnorm);
    hipLaunchKernelGGL(vq_prep_tokens_kernel, dim3((N + 255) / 256), dim3(256), 0, s, z_in, ldz, N, d, l2_normalized, zn, znorm, zz);
    const dim3 grid(nblk, S);
    const unsigned slo = (unsigned)(seed & 0xffffffffu), shi = (unsigned)(seed >> 32);
#define VQ_SEARCH(D2, M) hipLaunchKernelGGL((vq_search_kernel<D2, M>), grid, dim3(256), 0, s, zn, zz, ET, ee, N, K, Kp, inv_tau, slo, shi, cps, pscore, pidx, seed_counter)
#define VQ_SEARCH_D(M)                                   \
    switch (d) {                                         \
        case 8: VQ_SEARCH(4, M); break;                  \
        case 16: VQ_SEARCH(8, M); break;                 \
        case 24: VQ_SEARCH(12, M); break;                \
        default: VQ_SEARCH(16, M); break;                \
    }
    if (mode == 0) { VQ_SEARCH_D(0) } else if (mode == 1) { VQ_SEARCH_D(1) } else { VQ_SEARCH_D(2) }
#undef VQ_SEARCH_D
#undef VQ_SEARCH
    VT_CHECK_LAUNCH("vt_vq_forward/search");
    if (mode == 0)
        hipLaunchKernelGGL(vq_finalize_kernel<0>, dim3(nblk), dim3(256), 0, s, pscore, pidx, S, zn, E, N, d, idx, rz, (bf16_t*)rz_pad_bf16, ldp, partial);
    else
        hipLaunchKernelGGL(vq_finalize_kernel<1>, dim3(nblk), dim3(256), 0, s, pscore, pidx, S, zn, E, N, d, idx, rz, (bf16_t*)rz_pad_bf16, ldp, partial);
    hipLaunchKernelGGL(vq_loss_kernel, dim3(1), dim3(64), 0, s, partial, nblk, 1.0f / ((float)N * (float)d), beta, codebook_w, losses);
    VT_CHECK_LAUNCH("vt_vq_forward/finalize");
    return VT_OK;
}

extern "C" int vt_vq_forward(const float* z_in, int64_t ldz, const float* codebook, int32_t N, int32_t K, int32_t d, int32_t mode,
                             int32_t l2_normalized, float inv_tau, float beta, float codebook_w, uint64_t seed, float* E, float* wnorm,
                             float* zn, float* znorm, int64_t* idx, float* rz, void* rz_pad_bf16, int64_t ldp, float* losses,
                             void* workspace, vtStream stream) {
    return vt_vq_forward_ctr(z_in, ldz, codebook, N, K, d, mode, l2_normalized, inv_tau, beta, codebook_w, seed, nullptr, E, wnorm, zn, znorm, idx, rz,
                             rz_pad_bf16, ldp, losses, workspace, stream);
}

extern "C" int vt_vq_gather(const float* E, const int64_t* idx, int32_t N, int32_t K, int32_t d, float* out, void* out_pad_bf16,
                            int64_t ldp, vtStream stream) {
    VT_CHECK_ARG(E && idx && (out || out_pad_bf16) && N > 0 && K > 0 && d > 0, "vt_vq_gather: bad arguments");
    hipLaunchKernelGGL(vq_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, E, idx, N, K, d, out, (bf16_t*)out_pad_bf16, ldp);
    VT_CHECK_LAUNCH("vt_vq_gather");
    return VT_OK;
}

extern "C" int vt_vq_prep_codebook(const float* codebook, int32_t K, int32_t d, int32_t l2_normalized, float* E, float* wnorm,
                                   void* workspace, vtStream stream) {
    VT_CHECK_ARG(codebook && E && wnorm && workspace && K > 0 && d > 0, "vt_vq_prep_codebook: bad arguments");
    const int Kp = round_up(K, CHUNK);
    float* ET = (float*)workspace;
    float* ee = ET + (size_t)d * Kp;
    hipLaunchKernelGGL(vq_prep_codebook_kernel, dim3((Kp + 255) / 256), dim3(256), 0, (hipStream_t)stream, codebook, K, Kp, d, l2_normalized, E, ET, ee, wnorm);
    VT_CHECK_LAUNCH("vt_vq_prep_codebook");
    return VT_OK;
}

extern "C" int vt_vq_backward(const float* g_rz, int64_t ldg, const float* gscal, float beta, float codebook_w, const float* zn,
                              const float* znorm, const float* E, const float* wnorm, const int64_t* idx, int32_t N, int32_t K,
                              int32_t d, int32_t l2_normalized, float* dz_in, void* dz_pad_bf16, int64_t ldp, float* dW,
                              void* workspace, vtStream stream) {
    VT_CHECK_ARG(zn && znorm && E && wnorm && idx && (dz_in || dz_pad_bf16) && workspace, "vt_vq_backward: null pointer");
    VT_CHECK_ARG(N > 0 && K > 0 && (d == 8 || d == 16 || d == 24 || d == 32), "vt_vq_backward: d=%d must be 8,16,24 or 32", d);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(vq_bwd_tokens_kernel, dim3((N + 255) / 256), dim3(256), 0, s, g_rz, ldg, gscal, beta, codebook_w, zn, znorm, E, idx, N, d, l2_normalized, dz_in, (bf16_t*)dz_pad_bf16, ldp);
    if (!dW) {   // frozen codebook (the 'sq' quantizer, model_new/quantizer/fsq.py:165-167): no codebook gradient, 2NKd flops saved
        VT_CHECK_LAUNCH("vt_vq_backward");
        return VT_OK;
    }
    const int ns = cb_slabs(N);
    const int slab_len = round_up((N + ns - 1) / ns, CB_CHUNK);
    float* part = (float*)workspace;
    hipLaunchKernelGGL(vq_bwd_codebook_mfma_kernel, dim3((K + 127) / 128, ns), dim3(256), 0, s, zn, E, idx, N, K, d, slab_len, part);
    hipLaunchKernelGGL(vq_bwd_codebook_finalize, dim3((K + 7) / 8), dim3(256), 0, s, part, ns, gscal, codebook_w, E, wnorm, N, K, d, l2_normalized, dW);
    VT_CHECK_LAUNCH("vt_vq_backward");
    return VT_OK;
}
