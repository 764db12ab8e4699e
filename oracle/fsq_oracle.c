/* fsq_oracle.c -- TEST INFRASTRUCTURE ONLY: CPU restatement of the reference's finite scalar quantizer
 * (/root/reference/models/model_new/quantizer/fsq.py).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (libvt_hip.so) never does.
 *
 * Pinned by tests/golden/fsq_*.npz, which hold outputs of the reference class itself (imported by file path
 * in the build container, tests/golden/make_golden.py).
 *
 * Arithmetic, all in fp32 like the reference's autocast-disabled forward (fsq.py:119-131):
 *   half_l     = (levels - 1) * (1 + 1e-3) / 2                      fsq.py:78
 *   offset     = 0.5 where levels is even else 0                    fsq.py:79
 *   shift      = atanh(offset / half_l)                             fsq.py:80
 *   bounded    = tanh(z + shift) * half_l - offset                  fsq.py:81
 *   quantized  = round_half_even(bounded)        (straight-through) fsq.py:47-50,86
 *   codes      = quantized / (levels // 2)                          fsq.py:87-88
 *   indices    = int32( sum_c (codes_c * hw_c + hw_c) * basis_c )   fsq.py:90-92,103-107; basis = cumprod([1]+levels[:-1]) :67
 *   indices_to_codes: ((idx // basis) % levels - hw) / hw           fsq.py:94-101,109-113
 * tanh is evaluated in double and rounded to float (the correctly rounded fp32 value except in vanishingly rare
 * cases); the reference's vectorised fp32 tanh may differ from that by one ulp, which can move an element that
 * sits within an ulp of a rounding boundary -- the golden test skips exactly those elements and nothing else.
 */
#include <math.h>
#include <stdint.h>

#define FSQ_MAX_D 16

typedef struct {
    float half_l[FSQ_MAX_D], offset[FSQ_MAX_D], shift[FSQ_MAX_D], half_width[FSQ_MAX_D];
    int32_t basis[FSQ_MAX_D];
} fsq_consts;

void fsq_constants(const int32_t* levels, int d, float* half_l, float* offset, float* shift, float* half_width, int32_t* basis) {
    int32_t b = 1;
    for (int c = 0; c < d; ++c) {
        half_l[c] = (float)(levels[c] - 1) * (float)(1.0 + 1e-3) / 2.0f;
        offset[c] = (levels[c] % 2 == 0) ? 0.5f : 0.0f;
        const float ratio = offset[c] / half_l[c];
        shift[c] = (float)atanh((double)ratio);
        half_width[c] = (float)(levels[c] / 2);
        basis[c] = b;
        b *= levels[c];
    }
}

static void consts(const int32_t* levels, int d, fsq_consts* k) {
    fsq_constants(levels, d, k->half_l, k->offset, k->shift, k->half_width, k->basis);
}

/* z [N,d] -> codes [N,d], indices [N]; bounded (optional, [N,d]) exposes the pre-rounding value for near-tie masks */
void fsq_forward(const float* z, int64_t N, int d, const int32_t* levels, float* codes, int32_t* indices, float* bounded_out) {
    fsq_consts k;
    consts(levels, d, &k);
    for (int64_t n = 0; n < N; ++n) {
        float acc = 0.0f;
        for (int c = 0; c < d; ++c) {
            const float t = (float)tanh((double)(z[n * d + c] + k.shift[c]));
            const float bounded = t * k.half_l[c] - k.offset[c];
            const float q = rintf(bounded);
            const float code = q / k.half_width[c];
            codes[n * d + c] = code;
            if (bounded_out) bounded_out[n * d + c] = bounded;
            const float lvl = code * k.half_width[c] + k.half_width[c];
            acc = acc + lvl * (float)k.basis[c];
        }
        indices[n] = (int32_t)acc;
    }
}

/* what autograd derives from fsq.py:81-88 with the straight-through round: dz = (dcodes / hw) * half_l * (1 - tanh^2) */
void fsq_backward(const float* z, const float* dcodes, int64_t N, int d, const int32_t* levels, float* dz) {
    fsq_consts k;
    consts(levels, d, &k);
    for (int64_t n = 0; n < N; ++n)
        for (int c = 0; c < d; ++c) {
            const float t = (float)tanh((double)(z[n * d + c] + k.shift[c]));
            const float g = dcodes[n * d + c] / k.half_width[c];
            dz[n * d + c] = (g * k.half_l[c]) * (1.0f - t * t);
        }
}

void fsq_indices_to_codes(const int32_t* indices, int64_t N, int d, const int32_t* levels, float* codes) {
    fsq_consts k;
    consts(levels, d, &k);
    for (int64_t n = 0; n < N; ++n)
        for (int c = 0; c < d; ++c) {
            const int32_t lvl = (indices[n] / k.basis[c]) % levels[c];
            codes[n * d + c] = ((float)lvl - k.half_width[c]) / k.half_width[c];
        }
}
