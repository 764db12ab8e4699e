/* Plain-C restatement of the VQ nearest-codeword search with a DEFINED fp32 order.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/larp_oracle.py header).  Build: oracle/build.py
 * (gcc -O2 -ffp-contract=off -shared -fPIC).
 *
 * Restates /root/reference/models/bottleneck.py:
 *   F.normalize (eps 1e-12)                    :255, :267
 *   mode L  argmin(|z|^2 + |e|^2 - 2 z.e)      :282-290   (stochastic: false)
 *   mode D  argmax(softmax(z.e * 1/tau))       :275-278   (eval + set_eval_deterministic)
 *   q = E[idx]; regularized_z = z + (q - z)    :292-293, :307
 *   loss_commit / loss_codebook = mean((q-z)^2):295-298
 *
 * The reference leaves the accumulation order of the d-long dot product to the BLAS it
 * happens to run on.  This restatement FIXES it: every d-long reduction is a sequential
 * fp32 fused-multiply-add chain k = 0..d-1 starting from +0.0f.  That is exactly what a
 * chain of gfx950 v_mfma_f32_32x32x2_f32 instructions computes, so the HIP kernel can be
 * bit-identical to this file; agreement of this file with the reference's torch-CPU
 * results is measured (not assumed) by tests/test_oracle_golden.py on the fixtures in
 * tests/golden/ (identical except on sub-ulp near-ties, which the test counts).
 *
 * Mode D note: the reference takes argmax over softmax PROBABILITIES.  softmax is
 * monotone in the logit, so this file takes the first argmax of the fp32 logits
 * (dot * inv_tau); the two differ only if fp32 softmax rounds two different logits to
 * the same top probability, which needs a top-2 logit gap below one ulp of exp() ~6e-8
 * (never observed on the fixtures; the golden test would show it).
 */
#include <math.h>
#include <stdint.h>

static inline float dot_chain(const float *a, const float *b, int d) {
    float s = 0.0f;
    for (int k = 0; k < d; ++k) s = fmaf(a[k], b[k], s);
    return s;
}

/* out[r,:] = in[r,:] / max(sqrt(sum_k in[r,k]^2), eps);  norm_out[r] = that denominator */
void vq_normalize_rows(const float *in, int64_t rows, int d, int64_t in_stride,
                       float *out, float *norm_out, float eps) {
    for (int64_t r = 0; r < rows; ++r) {
        const float *x = in + r * in_stride;
        float s = dot_chain(x, x, d);
        float n = sqrtf(s);
        float den = n > eps ? n : eps;
        for (int k = 0; k < d; ++k) out[r * d + k] = x[k] / den;
        if (norm_out) norm_out[r] = den;
    }
}

/* mode 0 = L (argmin 3-term distance), 1 = D (argmax logit).  z, e already normalised
 * (or raw when l2_normalized is false).  Lowest index wins ties (torch.argmin/argmax). */
void vq_search(const float *z, const float *e, int64_t n, int64_t k_codes, int d, int mode,
               float inv_tau, int64_t *idx_out, float *score_out) {
    for (int64_t i = 0; i < n; ++i) {
        const float *zi = z + i * d;
        float zz = dot_chain(zi, zi, d);
        float best = 0.0f;
        int64_t bi = 0;
        for (int64_t c = 0; c < k_codes; ++c) {
            const float *ec = e + c * d;
            float dot = dot_chain(zi, ec, d);
            float sc;
            if (mode == 0) {
                float ee = dot_chain(ec, ec, d);
                float t = zz + ee;            /* fl(|z|^2 + |e|^2)            */
                sc = fmaf(-2.0f, dot, t);     /* fl(t - 2*dot); 2*dot is exact */
                if (c == 0 || sc < best) { best = sc; bi = c; }
            } else {
                sc = dot * inv_tau;
                if (c == 0 || sc > best) { best = sc; bi = c; }
            }
        }
        idx_out[i] = bi;
        if (score_out) score_out[i] = best;
    }
}

/* q = e[idx]; rz = z + (q - z); sq[i] = sum_k (q-z)^2 as a sequential chain (for the
 * loss; the reference's .mean() order is unspecified, so losses carry a tolerance). */
void vq_gather(const float *z, const float *e, const int64_t *idx, int64_t n, int d,
               float *q_out, float *rz_out, double *sq_sum_out) {
    double tot = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float *zi = z + i * d;
        const float *ec = e + idx[i] * d;
        for (int k = 0; k < d; ++k) {
            float diff = ec[k] - zi[k];
            q_out[i * d + k] = ec[k];
            rz_out[i * d + k] = zi[k] + diff;
            tot += (double)diff * (double)diff;
        }
    }
    *sq_sum_out = tot;
}

/* Dense codebook gradient in the summation order of the HIP path (vt_vq.hip: vq_bwd_codebook_mfma_kernel + _finalize):
 * autograd of F.embedding(sparse=False) through F.normalize (models/bottleneck.py:292-307):
 *   de[k] = s_b * sum_n [idx[n] == k] (e_k - z_n),   dW[k] = (de - e (e.de)) / |w_k|
 * with every slab of `slab_len` tokens summed as a sequential fp32 chain over ascending n (what the exact-fp32 MFMA
 * does with a one-hot operand), the slabs added in ascending order, e.de reduced by a 32-lane xor butterfly
 * (16, 8, 4, 2, 1) and de - e (e.de) formed by one fused multiply-add.  TEST INFRASTRUCTURE ONLY. */
void vq_codebook_grad(const float *zn, const float *e, const float *wnorm, const int64_t *idx, int64_t n, int64_t k_codes,
                      int d, float s_b, int normalize, int64_t slab_len, float *dW) {
    int64_t nslab = (n + slab_len - 1) / slab_len;
    for (int64_t k = 0; k < k_codes; ++k) {
        float de[32], lanes[32];
        for (int j = 0; j < 32; ++j) de[j] = 0.0f;
        for (int64_t s = 0; s < nslab; ++s) {
            float part[32];
            for (int j = 0; j < 32; ++j) part[j] = 0.0f;
            int64_t hi = (s + 1) * slab_len < n ? (s + 1) * slab_len : n;
            for (int64_t t = s * slab_len; t < hi; ++t)
                if (idx[t] == k)
                    for (int j = 0; j < d; ++j) {
                        volatile float diff = e[k * d + j] - zn[t * d + j];
                        part[j] = part[j] + diff;
                    }
            for (int j = 0; j < d; ++j) de[j] = de[j] + part[j];
        }
        for (int j = 0; j < 32; ++j) {
            de[j] = j < d ? de[j] * s_b : 0.0f;
            volatile float prod = j < d ? e[k * d + j] * de[j] : 0.0f;
            lanes[j] = prod;
        }
        for (int o = 16; o > 0; o >>= 1) {
            float nxt[32];
            for (int j = 0; j < 32; ++j) nxt[j] = lanes[j] + lanes[j ^ o];
            for (int j = 0; j < 32; ++j) lanes[j] = nxt[j];
        }
        for (int j = 0; j < d; ++j) {
            if (normalize) {
                dW[k * d + j] = fmaf(-e[k * d + j], lanes[j], de[j]) / wnorm[k];   /* one rounding, as the kernel's fmaf */
            } else {
                dW[k * d + j] = de[j];
            }
        }
    }
}
