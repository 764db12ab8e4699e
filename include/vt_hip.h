/* vt_hip.h -- C ABI of libvt_hip.so: the MI355X (gfx950) kernels behind the LARP tokenizer's
 * encode -> quantize -> decode training step.
 *
 * The reference (zhxie0117/video-tokenizer) is pure Python and has NO FFI of its own: the path
 * sits behind its model registry (models/models.py:8-27) and runs torch / timm library ops.
 * This header is the boundary the build introduces beneath those Python classes; each entry
 * point names the reference line whose arithmetic it replaces.  The Python side that binds
 * it (ctypes, tensor.data_ptr(), torch.cuda.current_stream().cuda_stream) is in
 * video-tokenizer_amd/hip.py; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, no exceptions.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - every function enqueues work on `stream` (a hipStream_t) and returns immediately;
 *     nothing here allocates, frees or synchronises (hipGraph-capturable).
 *   - the caller owns every buffer (outputs, saved-for-backward, workspaces); sizes come
 *     from the matching *_workspace_bytes().
 *   - return 0 on success, <0 on error; vt_last_error() gives the thread-local message.
 *   - "bf16" buffers are raw 16-bit bfloat16; `void*` is used for them in signatures.
 */
#ifndef VT_HIP_H
#define VT_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* vtStream; /* hipStream_t */

enum { VT_OK = 0, VT_ERR_INVALID = -1, VT_ERR_LAUNCH = -2, VT_ERR_UNSUPPORTED = -3 };

int vt_abi_version(void);
int vt_last_error(char* buf, size_t n);

/* row r of a logical [rows, dim] operand lives at physical row
 *   (r / grp) * stride + off + (r % grp)        (grp == 0: identity)
 * Used to read/write a slice "rows off..off+grp of every sequence of length stride" in place,
 * e.g. the last len(query) tokens that TransformerEncoderParallel keeps (transformer.py:69). */
typedef struct {
    int32_t grp;
    int64_t stride;
    int64_t off;
} vtRowMap;

/* ------------------------------------------------------------------------------------------
 * GEMM "NT": C[M,N] = A[M,K] . B[N,K]^T, bf16 operands, fp32 accumulate on MFMA.
 * Replaces nn.Linear / F.linear under autocast(bf16): timm Block's qkv/proj/fc1/fc2
 * (models/transformer.py:52-59), Bottleneck.in_linear/out_linear (models/bottleneck.py:109-110,
 * 142,163), OutputLayer.linear (models/larp_tokenizer.py:35,40) and the Conv3d of PatchEmbed3D
 * seen as a GEMM over gathered patches (models/embed.py:82,110), plus their input gradients.
 * K must be a multiple of 64; M and N are arbitrary (edge tiles are clamped on load, masked on
 * store).
 * ------------------------------------------------------------------------------------------ */
enum {
    VT_EPI_BF16 = 0,       /* out(bf16) = acc [+ bias]                                        */
    VT_EPI_BF16_GELU = 1,  /* out(bf16) = u = acc + bias ; out2(bf16) = gelu_erf(u)           */
    VT_EPI_F32 = 2,        /* out(f32)  = [round_bf16](acc [+ bias]) [+ residual] [+ rowmod]; */
                           /*             optional out2(bf16) copy; optional output row map   */
    VT_EPI_BF16_DGELU = 3  /* out(bf16) = acc * gelu_erf'(aux)      (aux = saved u, bf16)     */
};

typedef struct {
    const void* A; int64_t lda; /* bf16 [M,K] */
    const void* B; int64_t ldb; /* bf16 [N,K] */
    int32_t M, N, K;
    int32_t epi;
    void* out; int64_t ldo;
    void* out2; int64_t ldo2;
    const float* bias;                       /* [N] or NULL                                   */
    const float* residual; int64_t ldr;      /* f32, indexed like `out` (after the row map)   */
    const float* rowmod; int32_t rowmod_period; /* f32 [period,N] added at row (r % period)   */
    const void* aux; int64_t ldaux;          /* bf16 [M,N]                                    */
    vtRowMap omap;                           /* VT_EPI_F32 only                               */
    int32_t round_bf16;                      /* VT_EPI_F32 only                               */
    float* colsum_partial;                   /* VT_EPI_BF16_DGELU only, optional: fp32 [ceil(M/192), N]; row t = column
                                                sums of the bf16-rounded output over rows [192 t, 192 t + 192) (fixed
                                                order, no atomics).  vt_sum_slabs over the rows gives the bias gradient
                                                of the Linear whose pre-activation is `aux`, without re-reading `out` */
    float out_scale;                         /* VT_EPI_F32 only: 0 = none; otherwise the finished value (after bias, rounding, residual,
                                                rowmod) is multiplied by it -- the 1/sqrt(i+1) rescale of the residual stream after
                                                layer i of models/model_new/base/transformer.py:88-90 -- before out / out2 are written */
    int32_t tile;                            /* tile generation for THIS call: 0 = automatic (use this; 192x192x64 tiles on a
                                                3-stage LDS-DMA ring, one persistent workgroup per CU, when the problem fills
                                                them, else 128x128x64).  Tests / tuning: 1 = force 128, 2 = force 192, 5 = 192x96
                                                two workgroups per CU, 6 = 192x192 one tile per workgroup, 7 = the M <= 64
                                                weight-streaming kernel, 8..15 = 192x192 with every other CU starting 1..8 us late
                                                (timing experiment, correct results), 3/4/17/18 = timing ablations with WRONG results, 19 = 192x192 walking its
                                                tiles as a row-major list, 19 + W (W = 1..12) = in column blocks of W tile columns (automatic: 6 or 8 when N spans
                                                at least 6 tiles, else row-major; A/B timing of the order, same results).  Results do not depend on the choice (same fp32
                                                summation order) unless K is split, below.  The library keeps no such setting between calls */
    void* splitk_ws; int64_t splitk_ws_bytes; /* optional workspace of vt_gemm_nt_splitk_workspace_bytes() bytes, 256-byte aligned, ZERO when first
                                                handed over (its first 4 KiB are arrival counters the kernel returns to zero) and used by one stream
                                                at a time.  With it, a launch of the 128x128 kernel that would leave most CUs idle (the N = 768
                                                GEMMs of a one-clip training step: 72 tiles, K up to 3072) splits K over up to 8 workgroups per
                                                tile; the last to arrive adds the fp32 partial sums in split order and runs the epilogue:
                                                run-to-run bit-identical, but not the unsplit summation order.  NULL: never split */
    int32_t splitk;                          /* 0 = automatic (needs splitk_ws), 1 = never, 2..8 = this many (tests / tuning; tile 0, 1 or 16) */
} vtGemmNT;
size_t vt_gemm_nt_splitk_workspace_bytes(void);

int vt_gemm_nt(const vtGemmNT* p_host, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * GEMM "TN", grouped: for each problem g  C_g[P,Q] = A_g[M,P]^T . B_g[M,Q]  (fp32 out).
 * Weight gradients dW = dY^T X of every nn.Linear / the Conv3d above (what autograd derives
 * for F.linear); all problems of one launch share one grid so a transformer block's four
 * weight gradients fill the chip without split-K or atomics.  M % 64 == 0, P % 8 == 0,
 * Q % 8 == 0; rows >= p_lim and cols >= q_lim of C are not stored; row_perm (optional,
 * int32[p_lim]) scatters row p of C to row row_perm[p].  At most VT_TN_MAX_GROUP problems.
 * ------------------------------------------------------------------------------------------ */
#define VT_TN_MAX_GROUP 16
typedef struct {
    const void* A; int64_t lda; /* bf16 [M,P]  (dY) */
    const void* B; int64_t ldb; /* bf16 [M,Q]  (X)  */
    int32_t M, P, Q;
    float* out; int64_t ldo;
    int32_t p_lim, q_lim;
    const int32_t* row_perm;
    int32_t tile;               /* problem 0's value selects the tile generation of the launch: 0 = automatic, 1 = 128x128, 2 = 192x192,
                                   7 = 192x192 without the cross-K-tile register pipeline (round-1 kernel, A/B timing; same bits) */
} vtGemmTN;

int vt_gemm_tn_grouped(const vtGemmTN* problems_host, int32_t n_problems, vtStream stream);



/* ------------------------------------------------------------------------------------------
 * LayerNorm (fp32 in, fp32 statistics, bf16 out for the next MFMA GEMM).
 * Replaces nn.LayerNorm inside timm Block (norm1/norm2, eps 1e-5; models/transformer.py:52-59)
 * and OutputLayer.norm_final (eps 1e-6; models/larp_tokenizer.py:34,39), forward and backward.
 * dim in {256,512,768,1024}.  Row r of x/dres/dx/dx_bf16 is at physical row xmap(r); y, dy, mean,
 * rstd are indexed by r.  Backward also adds the residual-stream gradient `dres` (may be NULL)
 * and emits column sums: dgamma, dbeta and (optional) dxsum = sum_r dx[r,:] (the bias gradient
 * of the Linear that produced this residual stream).
 * ------------------------------------------------------------------------------------------ */
int vt_layernorm_fwd(const float* x, vtRowMap xmap, const float* gamma, const float* beta, float eps, int64_t rows,
                     int32_t dim, void* y_bf16, float* mean, float* rstd, vtStream stream);
size_t vt_layernorm_bwd_workspace_bytes(int32_t dim);
int vt_layernorm_bwd(const void* dy_bf16, const float* x, vtRowMap xmap, const float* gamma, const float* mean,
                     const float* rstd, const float* dres, int64_t rows, int32_t dim, float* dx, void* dx_bf16,
                     float* dgamma, float* dbeta, float* dxsum, void* workspace, vtStream stream);

/* column sums out[c] = sum_r src[map(r), c] (bias gradients of nn.Linear); src bf16 or fp32 */
size_t vt_colsum_workspace_bytes(int32_t width);
int vt_colsum(const void* src, int32_t src_is_bf16, int64_t ld, vtRowMap map, int64_t rows, int32_t width, float* out,
              void* workspace, vtStream stream);
/* out[j,:] = sum_b src[map(b*n + j), :]  -- gradient of encoder_latent_query_embed, which
 * larp_tokenizer.py:410 broadcasts over the batch with .repeat(b,1,1) */
int vt_batch_sum(const float* src, vtRowMap map, int32_t batch, int32_t n, int32_t dim, float* out, vtStream stream);
/* zero rows map(r), r < rows, of an fp32 matrix and/or its bf16 twin (rows a sliced backward does not write) */
int vt_zero_rows(float* a, void* b_bf16, vtRowMap map, int64_t rows, int32_t dim, vtStream stream);
/* out[c] = sum_s slabs[s*slab_stride + c], fixed order (combine split-M partial weight gradients) */
int vt_sum_slabs(const float* slabs, int32_t nslab, int64_t slab_stride, int32_t width, float* out, vtStream stream);
/* dst[r,:] = bf16(src[map(r),:]) */
int vt_cast_rows(const float* src, vtRowMap map, int64_t rows, int32_t dim, void* dst_bf16, int64_t ldd, vtStream stream);
/* dst[b*seq + off + j, :] = src[b*n + j, :] + table[j, :] + vec[:]   (each term optional)
 * sequence assembly: torch.cat([context, query]) of transformer.py:64 without a copy kernel per
 * operand, `z + decoder_latent_pe` (larp_tokenizer.py:463-464) and the query-embed broadcasts
 * (:410, :465). */
int vt_assemble_rows(float* dst, int64_t seq, int64_t off, int32_t batch, int32_t n, int32_t dim, const float* src,
                     const float* table, const float* vec, vtStream stream);
/* fp32 master weight W[N,K] -> bf16 W[N,ldd] and/or bf16 W^T[K,lddT]; packed row r = W[row_perm[r]] */
int vt_pack_weight(const float* w, int32_t N, int32_t K, const int32_t* row_perm, void* wb, int64_t ldd, void* wt,
                   int64_t lddT, vtStream stream);
/* the same for many weights at once (after an optimizer step every bf16 operand copy is refreshed); jobs is a host array */
#define VT_PACK_MAX_GROUP 32
typedef struct {
    const float* w; int32_t N, K; const int32_t* row_perm;
    void* wb; int64_t ldd; void* wt; int64_t lddT;
} vtPackJob;
int vt_pack_weights_grouped(const vtPackJob* jobs_host, int32_t n_jobs, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * (T,H,W) patchify / unpatchify.  Patch-row columns are in (c,dt,dy,dx) order == Conv3d weight
 * flattening (models/embed.py:82,110-112); unpatchify is the inverse and, with the head weight
 * rows permuted at pack time, replaces LARPTokenizer.unpatchify's einops permute + .contiguous()
 * (models/larp_tokenizer.py:441-454,493).  p % 8 == 0.
 * ------------------------------------------------------------------------------------------ */
int vt_patchify(const float* video, int32_t B, int32_t C, int32_t T, int32_t S, int32_t pt, int32_t p, void* rows_bf16,
                vtStream stream);
int vt_unpatchify(const float* rows, int32_t B, int32_t C, int32_t T, int32_t S, int32_t pt, int32_t p, float* video,
                  vtStream stream);

/* ------------------------------------------------------------------------------------------
 * Fused attention, head_dim 64, no mask (timm Attention -> F.scaled_dot_product_attention inside
 * timm Block, models/transformer.py:52-59).  qkv bf16 [B,L,3,H,64]; o/dO bf16 [B,L,H,64];
 * lse2 fp32 [B,H,L] (log2-domain LSE of the scaled scores, saved for backward);
 * delta_ws fp32 [B,H,L] scratch.  Any L >= 1 (tail rows/keys are masked).  qkv, o, dO and dqkv 16-byte aligned (every
 * operand is read and every output written 16 bytes per lane; a misaligned pointer returns VT_ERR_INVALID).
 * ------------------------------------------------------------------------------------------ */
int vt_attention_fwd(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, void* o, float* lse2, vtStream stream);
int vt_attention_bwd(const void* qkv, const void* o, const void* dO, const float* lse2, int32_t B, int32_t L, int32_t H,
                     int32_t hd, void* dqkv, float* delta_ws, vtStream stream);
/* causal variant (head_dim 64): query q attends keys 0..q -- F.scaled_dot_product_attention(..., is_causal=True) inside the AR
 * consumer's Attention (models/larp_ar.py:186-190) and its autograd.  Same layouts as above. */
int vt_attention_causal_fwd(const void* qkv, int32_t B, int32_t L, int32_t H, void* o, float* lse2, vtStream stream);
int vt_attention_causal_bwd(const void* qkv, const void* o, const void* dO, const float* lse2, int32_t B, int32_t L, int32_t H, void* dqkv,
                            float* delta_ws, vtStream stream);
/* The same for the queries q_begin .. L-1 only (q_begin a multiple of 64): `transformer_encoder_parallel` returns
 * h[:, -len(query):] (models/transformer.py:69), so in the LAST block of a stack the other rows' attention outputs are
 * never read and their output gradients are zero.  o / dO are compact [B, L - q_begin, H, hd]; keys/values, lse2,
 * delta_ws and dqkv keep the full length (dQ rows before q_begin are written as zeros, dK/dV get the kept queries'
 * contributions). */
int vt_attention_fwd_rows(const void* qkv, int32_t B, int32_t L, int32_t H, int32_t hd, int32_t q_begin, void* o_compact, float* lse2,
                          vtStream stream);
int vt_attention_bwd_rows(const void* qkv, const void* o_compact, const void* dO_compact, const float* lse2, int32_t B, int32_t L,
                          int32_t H, int32_t hd, int32_t q_begin, void* dqkv, float* delta_ws, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * Vector quantisation (SimpleVectorQuantizer.forward, models/bottleneck.py:262-324).
 *   mode 0: stochastic=False  argmin(|z|^2+|e|^2-2z.e)           (:282-290)
 *   mode 1: eval-deterministic argmax(softmax(z.e/tau))           (:275-278)
 *   mode 2: training default   multinomial(softmax(z.e/tau))      (:280)  via Gumbel-max, counter RNG
 * z_in fp32 [N,ldz]; codebook fp32 [K,d] (nn.Embedding weight); d in {8,16,24,32}.
 * Outputs (all caller-owned, saved for backward): E [K,d] = normalised codebook ('emb'), wnorm [K],
 * zn [N,d] ('unregularized_z'), znorm [N], idx int64 [N] ('bottleneck_rep'), rz fp32 [N,d]
 * ('regularized_z' = z + (q - z)), optional rz_pad bf16 [N,ldp] (cols >= d untouched),
 * losses[4] = {loss_q, loss_commit, loss_codebook, mse}.  Index arithmetic is bit-exact against
 * oracle/vq_oracle.c (sequential fp32 FMA chain; lowest index wins ties).
 * Backward: gscal = device {dL/dloss_q, dL/dloss_commit, dL/dloss_codebook} (may be NULL);
 * g_rz = dL/dregularized_z fp32 [N,ldg] (may be NULL); outputs dz_in (fp32 [N,d] and/or bf16
 * [N,ldp]) and the dense codebook gradient dW [K,d] (deterministic, no atomics: one-hot product on the exact
 * fp32 MFMA, cost independent of how the tokens spread over the codes); dW == NULL skips it (frozen codebook: the 'sq'
 * quantizer `VectorQuantizer` of models/model_new/quantizer/fsq.py:144-230, K = 196 560, which is vt_vq_forward mode 1 with
 * inv_tau = 1 and loss = d * loss_q).  `workspace`: vt_vq_workspace_bytes(N,K,d) bytes, the forward's scratch may be reused.
 * ------------------------------------------------------------------------------------------ */
size_t vt_vq_workspace_bytes(int32_t N, int32_t K, int32_t d);
int vt_vq_forward(const float* z_in, int64_t ldz, const float* codebook, int32_t N, int32_t K, int32_t d, int32_t mode,
                  int32_t l2_normalized, float inv_tau, float beta, float codebook_w, uint64_t seed, float* E, float* wnorm,
                  float* zn, float* znorm, int64_t* idx, float* rz, void* rz_pad_bf16, int64_t ldp, float* losses,
                  void* workspace, vtStream stream);
/* the same with a per-call counter that lives in DEVICE memory (mode 2 only; may be NULL): the sampling noise is hashed from
 * seed + *seed_counter, so a captured launch draws fresh noise on every replay once the caller increments the counter in the graph */
int vt_vq_forward_ctr(const float* z_in, int64_t ldz, const float* codebook, int32_t N, int32_t K, int32_t d, int32_t mode,
                      int32_t l2_normalized, float inv_tau, float beta, float codebook_w, uint64_t seed, const uint32_t* seed_counter,
                      float* E, float* wnorm, float* zn, float* znorm, int64_t* idx, float* rz, void* rz_pad_bf16, int64_t ldp,
                      float* losses, void* workspace, vtStream stream);
int vt_vq_backward(const float* g_rz, int64_t ldg, const float* gscal, float beta, float codebook_w, const float* zn,
                   const float* znorm, const float* E, const float* wnorm, const int64_t* idx, int32_t N, int32_t K,
                   int32_t d, int32_t l2_normalized, float* dz_in, void* dz_pad_bf16, int64_t ldp, float* dW,
                   void* workspace, vtStream stream);
/* get_codebook_entry (bottleneck.py:327-344): E = normalise(codebook) then out[n,:] = E[idx[n],:] */
int vt_vq_prep_codebook(const float* codebook, int32_t K, int32_t d, int32_t l2_normalized, float* E, float* wnorm,
                        void* workspace, vtStream stream);
int vt_vq_gather(const float* E, const int64_t* idx, int32_t N, int32_t K, int32_t d, float* out, void* out_pad_bf16,
                 int64_t ldp, vtStream stream);


/* ------------------------------------------------------------------------------------------
 * Finite scalar quantizer: FSQ.forward / quantize / codes_to_indices / indices_to_codes of the reference's
 * TiTok-style autoencoders (models/model_new/quantizer/fsq.py:76-131; levels [8,8,8,5,5,5] or [8,8,8,8,5,5,5,5],
 * models/model_new/autoencoder.py:59,140,640).  z, codes, dcodes, dz are [N, d] row-major, fp32 (is_bf16 = 0) or
 * bf16 (is_bf16 = 1: the reference up-casts to fp32, computes, and casts codes back, :122-129); indices int32 [N]
 * (may be NULL in forward).  `levels_host`: d ints on the HOST, each >= 2, prod <= 2^24, d <= 16.
 * Backward is the straight-through estimator autograd derives: dz = dcodes / (levels // 2) * half_l * (1 - tanh^2).
 * One launch each; indices and codes are bit-exact against the reference's fp32 arithmetic.
 * ------------------------------------------------------------------------------------------ */
int vt_fsq_codebook_size(const int32_t* levels_host, int32_t d, int64_t* size);
int vt_fsq_forward(const void* z, int32_t is_bf16, int64_t N, int32_t d, const int32_t* levels_host, void* codes,
                   int32_t* indices, vtStream stream);
int vt_fsq_backward(const void* z, const void* dcodes, int32_t is_bf16, int64_t N, int32_t d, const int32_t* levels_host,
                    void* dz, vtStream stream);
int vt_fsq_indices_to_codes(const int32_t* indices, int64_t N, int32_t d, const int32_t* levels_host, void* codes,
                            int32_t is_bf16, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * Glue of the TiTok-style transformer block (models/model_new/base/transformer.py:11-63, rope.py:18-24): the block's
 * GEMMs, its LayerNorm over D and the attention itself are vt_gemm_nt / vt_layernorm_* / vt_attention_*; these are the
 * HBM-bound passes in between.  All matrices bf16 row-major, 16-byte aligned; head_dim is 64 (utils.py:6), D = 64 H.
 *   qkvg [M, 4D]  = to_qkv(x), columns q | k | v | gate (chunk(4), :46), M = B * L rows, row r has position r % L.
 *   vt_qknorm_rope_fwd: q, k <- rotary(LayerNorm_64(q or k; eps, affine q_w/q_b, k_w/k_b)) (:52-56), v copied; writes
 *     the packed [M, 3D] operand of vt_attention_fwd.  cos/sin: fp32 [L, 32] = real/imag of freqs_cis (rope.py:27-46,
 *     95-112), pair j of a head rotates (x[2j], x[2j+1]).
 *   vt_qknorm_rope_bwd: dqkv [M, 3D] (from vt_attention_bwd) -> columns 0..3D of dqkvg [M, 4D]; parameter gradients
 *     dq_w, dq_b, dk_w, dk_b fp32 [64] each (any may be NULL); workspace of vt_qknorm_rope_bwd_workspace_bytes().
 *   vt_sigmoid_gate_fwd: og [M, D] = o * sigmoid(gate) (:61), gate read in place from columns 3D..4D of qkvg.
 *   vt_sigmoid_gate_bwd: dog -> d_o [M, D] and columns 3D..4D of dqkvg.
 *   vt_geglu_fwd: a[:, :I] = gelu_erf(h[:, I:2I]) * h[:, :I]  (GEGLU, :11-17), h [M, 2I], a has row stride lda >= I
 *     (pad columns are not written: zero them once if lda is the 64-padded contraction dim of the next GEMM).
 *   vt_geglu_bwd: da -> dh [M, 2I].
 * Rounding follows autocast(bf16): every tensor the reference materialises in bf16 is rounded at the same point.
 * ------------------------------------------------------------------------------------------ */
int vt_qknorm_rope_fwd(const void* qkvg, int64_t M, int32_t L, int32_t H, const float* q_w, const float* q_b, const float* k_w,
                       const float* k_b, float eps, const float* cos_tab, const float* sin_tab, void* qkv_out, vtStream stream);
size_t vt_qknorm_rope_bwd_workspace_bytes(void);
int vt_qknorm_rope_bwd(const void* qkvg, const void* dqkv, int64_t M, int32_t L, int32_t H, const float* q_w, const float* k_w,
                       float eps, const float* cos_tab, const float* sin_tab, void* dqkvg, float* dq_w, float* dq_b, float* dk_w,
                       float* dk_b, void* workspace, vtStream stream);
int vt_sigmoid_gate_fwd(const void* o, const void* qkvg, int64_t M, int32_t D, void* og, vtStream stream);
int vt_sigmoid_gate_bwd(const void* dog, const void* o, const void* qkvg, int64_t M, int32_t D, void* d_o, void* dqkvg, vtStream stream);
int vt_geglu_fwd(const void* h, int64_t M, int32_t I, void* a, int64_t lda, vtStream stream);
int vt_geglu_bwd(const void* da, int64_t lda, const void* h, int64_t M, int32_t I, void* dh, vtStream stream);
/* dst[r, :] = scale * src[r, :] (fp32, may be in place) and/or its bf16 copy: the incoming gradient of a layer whose output was
 * rescaled by 1/sqrt(i+1), produced once for the fp32 residual path and the bf16 GEMM operand.  dim % 4 == 0. */
int vt_scale_rows(const float* src, float scale, int64_t rows, int32_t dim, float* dst_f32, void* dst_bf16, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * AR consumer (`LARP_AR`, models/larp_ar.py:233-438; consumes the tokenizer's `bottleneck_rep`): the kernels its layers need
 * beyond vt_gemm_* and vt_attention_causal_*.
 *   vt_rmsnorm_fwd / _bwd: RMSNorm (models/norm.py:6-17) y = x * rsqrt(mean(x^2) + eps) * w, x fp32 [rows, dim], y bf16, rstd fp32
 *     [rows] saved; backward adds `dres` (may be NULL), writes dx fp32 and/or bf16 and dw [dim] (fixed-order partial sums in
 *     `workspace` of vt_rmsnorm_bwd_workspace_bytes(dim)).  dim in {384, 768, 1024, 1280, 1536, 2560} (the llama-abs sizes).
 *   vt_swiglu_fwd / _bwd: FeedForward (larp_ar.py:122-136) a = silu(w1 x) * (w3 x) on the packed projection h [M, 2I] = [w3 x | w1 x].
 *   vt_decode_attention: one new token per sequence against the KV cache (larp_ar.py:138-190, `mask = causal_mask[:, None, pos]`):
 *     q bf16 [B, H, 64], caches bf16 [>=B, H, Lmax, 64], keys 0..n_keys-1 -> o bf16 [B, H, 64].
 *   vt_decode_attention_step: the same for the generation loop (ar/generate.py:99-123) without a host round trip: `pos_dev` (device
 *     int32: the new token's position = number of earlier keys) is read by the kernel, the new token's k / v are taken from the
 *     packed projection row qkv [B, 3 * H * 64] = [q | k | v], stored into the caches at `pos` (KVCache.update, larp_ar.py:153-161)
 *     and attended together with keys 0..pos-1.  With it one decode step has no host-side dependency and can be replayed as a hipGraph.
 * ------------------------------------------------------------------------------------------ */
int vt_rmsnorm_fwd(const float* x, const float* w, float eps, int64_t rows, int32_t dim, void* y_bf16, float* rstd, vtStream stream);
size_t vt_rmsnorm_bwd_workspace_bytes(int32_t dim);
int vt_rmsnorm_bwd(const void* dy_bf16, const float* x, const float* w, const float* rstd, const float* dres, int64_t rows, int32_t dim,
                   float* dx, void* dx_bf16, float* dw, void* workspace, vtStream stream);
int vt_swiglu_fwd(const void* h, int64_t M, int32_t I, void* a, vtStream stream);
int vt_swiglu_bwd(const void* da, const void* h, int64_t M, int32_t I, void* dh, vtStream stream);
int vt_decode_attention(const void* q, const void* k_cache, const void* v_cache, int32_t B, int32_t H, int64_t Lmax, int32_t n_keys, void* o,
                        vtStream stream);
int vt_decode_attention_step(const void* qkv, void* k_cache, void* v_cache, int32_t B, int32_t H, int64_t Lmax, const int32_t* pos_dev, void* o,
                             vtStream stream);
/* vt_decode_norm_linear: Linear(RMSNorm(x)) for the M <= 64 token rows of a decode step as ONE launch (x fp32 [M, K], norm_w fp32 [K],
 * W bf16 [N, K] row-major, K % 128 == 0, N % 16 == 0).  mode 0: out bf16 [M, ldo >= N] (wqkv); mode 1: W = [w3 ; w1] interleaved in slabs
 * of 8 + 8 rows, out bf16 [M, ldo >= N / 2] = silu(w1 y) * (w3 y) (FeedForward, larp_ar.py:135); mode 2: out fp32 [M, ldo >= N] holding
 * the bf16-rounded values (the LM head).  Bit-identical to vt_rmsnorm_fwd + vt_gemm_nt(tile 7) [+ vt_swiglu_fwd].
 * norm_w == NULL: x is the already normalised bf16 [M, K] operand (only the Linear and its epilogue are fused). */
int vt_decode_norm_linear(const void* x, const float* norm_w, float eps, const void* W_bf16, int32_t M, int32_t N, int32_t K, int32_t mode, void* out,
                          int64_t ldo, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * A stack of those layers as ONE enqueue per direction: `ResidualAttentionBlock.forward` (transformer.py:66-91) and the
 * backward autograd derives for it -- x <- (x + Attn(x)); x <- (x + ffd(x)); x <- x / sqrt(i+1) for i = 0..depth-1 -- the
 * transformer of every `autoencoder_*` model (Encoder / Decoder / Decoder_unify, base/blocks.py).  Per layer forward: cast,
 * 4 GEMMs, vt_qknorm_rope_fwd, vt_attention_fwd, vt_sigmoid_gate_fwd, vt_layernorm_fwd, vt_geglu_fwd (11 launches); backward
 * 16 launches + one grouped weight-gradient launch.  The caller owns parameters / gradients (fp32, the reference's layout:
 * to_qkv [4D, D], q/k_norm [64], out_proj [D, D], ffd LayerNorm [D], ffd.1 [2 inner, D], ffd.3 [D, inner]), the rotary tables
 * (fp32 [L, 32] each) and one workspace per forward whose backward is pending.  B * L % 64 == 0, D = 64 H.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t B, L, D, H, depth, inner; } vtGatedStackConfig;
typedef struct { float *to_qkv_w, *q_norm_w, *q_norm_b, *k_norm_w, *k_norm_b, *out_proj_w, *ln_w, *ln_b, *fc1_w, *fc2_w; } vtGatedLayerTensors;
typedef struct vtGatedStack vtGatedStack;
int vt_gated_stack_create(const vtGatedStackConfig* cfg, vtGatedStack** out);
void vt_gated_stack_destroy(vtGatedStack* st);
size_t vt_gated_stack_workspace_bytes(const vtGatedStack* st);
int vt_gated_stack_init_workspace(vtGatedStack* st, void* ws, vtStream stream);
/* repack != 0: refresh the bf16 operand copies of the weights first (after an optimizer step / load_state_dict) */
int vt_gated_stack_forward(vtGatedStack* st, const vtGatedLayerTensors* layers_host, const float* cos_tab, const float* sin_tab,
                           const float* x_in, void* ws, float* x_out, int32_t repack, vtStream stream);
int vt_gated_stack_backward(vtGatedStack* st, const vtGatedLayerTensors* layers_host, const float* cos_tab, const float* sin_tab,
                            const float* dy, void* ws, const vtGatedLayerTensors* grads_host, float* dx, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * Fused Adam (+ optional EMA) over flat fp32 buffers: replaces optimizer.step() of torch.optim.Adam
 * (trainers/larp_tokenizer_trainer.py:160-212 builds Adam(lr, betas); :376-377 steps it) and
 * update_ema (trainers/base_trainer.py:769-779), one HBM-bound pass.  n % 4 == 0; step counts from 1;
 * ema may be NULL.  torch.optim.Adam semantics (L2 weight decay, no amsgrad).
 * ------------------------------------------------------------------------------------------ */
int vt_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, int32_t step, float* ema, float ema_decay, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * Whole-model engine: LARPTokenizer.forward = encode -> bottleneck(VQ) -> decode
 * (models/larp_tokenizer.py:400-428, 456-469, 489-496; TransformerEncoderParallel.forward
 * models/transformer.py:62-70; Bottleneck.forward models/bottleneck.py:170-188) and the backward
 * autograd derives for it, as ONE enqueue per direction (no per-op host work; hipGraph-capturable).
 * Mixed precision follows torch.autocast(bf16) on the reference: fp32 parameters, residual stream,
 * LayerNorm statistics, VQ and pixels; bf16 MFMA operands with fp32 accumulation.
 *
 * The caller owns: parameters/gradients (fp32, reference state-dict layout), one workspace of
 * vt_tokenizer_workspace_bytes() (zeroed once by vt_tokenizer_init_workspace), and the outputs.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t B, C, T, S, pt, p;            /* batch per GPU, channels, frames, side, patch sizes      */
    int32_t D, H, depth_enc, depth_dec;   /* hidden (768), heads (12 => head_dim 64), block counts   */
    int32_t Nq, d, K;                     /* bottleneck_token_num, bottleneck_dim, codebook_size     */
    int32_t vq_mode, l2_normalized;       /* 0 l2-argmin | 1 cos-argmax | 2 cos-sample               */
    float inv_tau, beta, codebook_w;      /* 1/stochastic_temperature, commitment / codebook weights */
    int32_t freeze_codebook;              /* 1: no codebook gradient is computed or written (bottleneck_type 'sq',
                                             models/larp_tokenizer.py:225-229: frozen 196 560 x 24 codebook) */
} vtTokenizerConfig;

typedef struct {                          /* one timm Block; same struct (non-const use) for grads   */
    float *norm1_w, *norm1_b, *qkv_w, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} vtBlockTensors;

typedef struct {                          /* parameters OR their gradients (buffers NULL in grads)   */
    float *pe_w, *pe_b;                   /* x_embedder.proj.{weight [D,C*pt*p*p], bias}             */
    float *enc_patch_pe;                  /* encoder_patch_pe [Nv,D]           (buffer)              */
    float *enc_query;                     /* encoder_latent_query_embed [Nq,D]                        */
    float *dec_latent_pe;                 /* decoder_latent_pe [Nq,D]          (buffer)              */
    float *dec_patch_query;               /* decoder_patch_query_embed [Nv,D]  (buffer)              */
    float *dec_token_type;                /* decoder_patch_query_token_type_embed [D] or NULL         */
    float *in_w, *in_b, *out_w, *out_b;   /* bottleneck.in_linear / out_linear                        */
    float *codebook;                      /* bottleneck.regularizer.embedding.weight [K,d]            */
    float *head_norm_w, *head_norm_b, *head_w, *head_b; /* final_layer.norm_final / linear          */
    const vtBlockTensors* enc_blocks;     /* host array [depth_enc]                                   */
    const vtBlockTensors* dec_blocks;     /* host array [depth_dec]                                   */
} vtTokenizerTensors;

typedef struct {                          /* forward outputs (device, caller-owned)                   */
    float* pred_frames;                   /* [B,C,T,S,S]                                              */
    float* encoded;                       /* [B,Nq,D]  'encoded'                                      */
    int64_t* indices;                     /* [B,Nq]    'bottleneck_rep'                               */
    float* projected_z;                   /* [B*Nq,d]  'projected_z' (in_linear output)               */
    float* unregularized_z;               /* [B*Nq,d]                                                 */
    float* regularized_z;                 /* [B*Nq,d]                                                 */
    float* emb;                           /* [K,d]     'emb' (normalised codebook)                    */
    float* losses;                        /* [4] loss_q, loss_commit, loss_codebook, mse              */
    float* input_norms;                   /* [2] input_norm_first, input_norm_last                    */
} vtTokenizerOutputs;

typedef struct vtTokenizer vtTokenizer;

int vt_tokenizer_create(const vtTokenizerConfig* cfg, vtTokenizer** out);
void vt_tokenizer_destroy(vtTokenizer* tk);
size_t vt_tokenizer_workspace_bytes(const vtTokenizer* tk);
int vt_tokenizer_init_workspace(vtTokenizer* tk, void* workspace, vtStream stream);
/* fp32 master weights -> packed bf16 operands (call after every optimizer step) */
int vt_tokenizer_pack(vtTokenizer* tk, const vtTokenizerTensors* params, void* workspace, vtStream stream);
/* encode: video -> encoded + VQ outputs.  decode: encoded -> pred_frames.  forward = both. */
int vt_tokenizer_encode(vtTokenizer* tk, const vtTokenizerTensors* params, const float* video, void* workspace,
                        const vtTokenizerOutputs* out, uint64_t seed, vtStream stream);
int vt_tokenizer_decode(vtTokenizer* tk, const vtTokenizerTensors* params, const float* encoded, void* workspace,
                        float* pred_frames, vtStream stream);
/* bottleneck.decode (models/bottleneck.py:166-168): indices -> encoded */
int vt_tokenizer_codes_to_encoded(vtTokenizer* tk, const vtTokenizerTensors* params, const int64_t* indices, void* workspace,
                                  float* encoded, vtStream stream);
/* backward of forward(); stages run in order head(0), decoder blocks (1..depth_dec, last block
 * first), bottleneck, encoder blocks, patch-embed; [stage_begin, stage_end) lets the caller
 * interleave gradient all-reduce buckets between stages.  d_pred [B,C,T,S,S] fp32;
 * gscal = device {dL/dloss_q, dL/dloss_commit, dL/dloss_codebook} or NULL.
 * The weight-gradient GEMMs of up to 4 consecutive transformer blocks are deferred and issued as ONE
 * grouped launch (768 tiles of 192x192 = 3 full rounds of the 256 CUs), so the gradients of a stage
 * may become final a few stages later: *final_through (optional) receives the number of leading stages
 * whose gradients are complete once the enqueued work has run. */
int32_t vt_tokenizer_num_backward_stages(const vtTokenizer* tk);
/* device-side per-call counter of the stochastic quantizer (see vt_vq_forward_ctr); NULL (the default) = the by-value seed alone */
int vt_tokenizer_set_seed_counter(vtTokenizer* tk, const uint32_t* seed_counter);
/* Split K in the BACKWARD input-gradient GEMMs that would leave most CUs idle (vtGemmNT.splitk_ws; one or two clips per GPU): on by
 * default (VT_GEMM_SPLITK=0 in the environment starts every handle with it off).  Forward GEMMs are never split: a clip's tokens and
 * reconstruction do not depend on the batch it runs in.  With the split, its GRADIENTS are summed in a different fp32 order than in a
 * larger batch and, re-rounded to bf16 layer after layer, differ from them at the bf16 noise level (~5e-3 relative on the deepest
 * layers) instead of ~1e-6: switch it off where bit-stable gradients across batch sizes matter more than the ~8 % it buys. */
int vt_tokenizer_set_split_k(vtTokenizer* tk, int32_t on);
/* Data-parallel runs (ABI 7).  The weight gradients of four consecutive blocks are produced by ONE grouped launch, so a group's gradients are
 * final -- and DistributedDataParallel's bucketed all-reduce of them (trainers/base_trainer.py:388) can start -- only there; by default the
 * encoder's blocks 3..0 finish at the very end of backward and their 113 MB (config B) are reduced with nothing left to overlap.  n > 0: the
 * encoder's blocks below n flush block by block (n = 3: groups 3-2 | 1 | 0), leaving one block's gradients for the exposed tail at the price
 * of three launches that do not fill whole rounds of the chip.  Bit-identical gradients.  parallel.DataParallelTokenizer sets n = 3. */
int vt_tokenizer_set_wgrad_tail(vtTokenizer* tk, int32_t n);
/* Data-parallel runs (ABI 7).  side != NULL: the grouped weight-gradient launches and the partial-sum reductions of the same blocks are
 * enqueued on `side` instead of the caller's stream, behind an event of the caller's stream; the caller's stream waits for them at the last
 * stage of backward (and before it rewrites an operand buffer a pending group still reads).  They are off the critical path of backward, and
 * while a collective's workgroups hold CUs the exact-fit GEMM launches of the main stream leave most of the chip idle in their extra round:
 * an independent stream fills it.  A consumer of a finished gradient slice (final_through of vt_tokenizer_backward) must wait for BOTH
 * streams.  Same kernels on the same operands: bit-identical gradients.  NULL restores the single-stream schedule.  Not capturable. */
int vt_tokenizer_set_wgrad_stream(vtTokenizer* tk, vtStream side);
/* Blocks per grouped weight-gradient launch, 1..4 (default 4 = 768 tiles = three whole rounds of the chip at config B).  (ABI 7) */
int vt_tokenizer_set_wgrad_batch(vtTokenizer* tk, int32_t n);
int vt_tokenizer_backward(vtTokenizer* tk, const vtTokenizerTensors* params, const float* d_pred, const float* gscal,
                          void* workspace, const vtTokenizerTensors* grads, int32_t stage_begin, int32_t stage_end,
                          int32_t* final_through, vtStream stream);
/* Data-parallel runs (ABI 8).  on != 0: a collective's workgroups share the chip with the backward (DistributedDataParallel's bucketed
 * all-reduce, trainers/base_trainer.py:388), so the backward's GEMMs with more 192x192 tiles than CUs are launched one tile per workgroup
 * (vtGemmNT.tile = 6) and the hardware hands tiles to whichever CU is free.  Bit-identical gradients.  (ABI 7 tied this to
 * vt_tokenizer_set_wgrad_tail(n > 0).) */
int vt_tokenizer_set_data_parallel(vtTokenizer* tk, int32_t on);
/* vt_tokenizer_backward for a caller that reduces finished gradient slices between stages (ABI 8): runs stages from stage_begin until
 * *final_through advances (a group of weight gradients was flushed) or the last stage is done; *stage_next = the stage to go on with
 * (== vt_tokenizer_num_backward_stages() when the backward is complete).  ~8 calls per backward at 12 + 12 blocks instead of 27. */
int vt_tokenizer_backward_until_flush(vtTokenizer* tk, const vtTokenizerTensors* params, const float* d_pred, const float* gscal,
                                      void* workspace, const vtTokenizerTensors* grads, int32_t stage_begin, int32_t* stage_next,
                                      int32_t* final_through, vtStream stream);

/* ------------------------------------------------------------------------------------------
 * A stack of timm Blocks on its own (fp32 [B, L, D] in and out): `transformer_encoder_parallel` / `_fused` called
 * outside LARPTokenizer (models/transformer.py:8-70; cat / slice of context and query stay with the caller) and the
 * discriminator's encoder (models/loss.py:150-155: width 384, 12 heads of 32, L = 1025).  head_dim 64 or 32; D in
 * {128,256,384,512,768,1024}; any L.  The forward re-packs the bf16 operand copies of the weights, saves its
 * activations in `ws` (vt_stack_workspace_bytes, zeroed once by vt_stack_init_workspace -- padded rows must stay
 * zero) and the backward of the SAME ws returns dL/dx and, unless need_wgrad == 0 (frozen stack: input gradient
 * only), every parameter gradient (weight gradients of 4 blocks per grouped launch).  A second forward before
 * the first one's backward needs its own ws.
 * ------------------------------------------------------------------------------------------ */
typedef struct { int32_t B, L, D, H, depth; } vtStackConfig;
typedef struct vtTokenizer vtStack;
int vt_stack_create(const vtStackConfig* cfg, vtStack** out);
void vt_stack_destroy(vtStack* st);
size_t vt_stack_workspace_bytes(const vtStack* st);
int vt_stack_init_workspace(vtStack* st, void* ws, vtStream stream);
int vt_stack_set_split_k(vtStack* st, int32_t on);   /* as vt_tokenizer_set_split_k */
int vt_stack_forward(vtStack* st, const vtBlockTensors* blocks_host, const float* x_in, void* ws, float* x_out, vtStream stream);
int vt_stack_backward(vtStack* st, const vtBlockTensors* blocks_host, const float* dy, void* ws, const vtBlockTensors* grads_host,
                      float* dx, int32_t need_wgrad, vtStream stream);

#ifdef __cplusplus
}
#endif
#endif /* VT_HIP_H */
