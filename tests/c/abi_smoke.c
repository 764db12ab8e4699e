/* Plain-C client of include/vt_hip.h: proves the boundary is a C ABI (no C++/torch types), that the header compiles as
 * C, and that argument validation reports through the return code + vt_last_error without touching a GPU. */
#include <stdio.h>
#include <string.h>

#include "vt_hip.h"

/* The call sequence of INTEGRATION.md section 3, compile-checked against the header (never executed here: it needs a GPU). */
static int example_training_step(vtTokenizer* tk, const vtTokenizerTensors* params, const vtTokenizerTensors* grads, const float* video,
                                 void* ws, const vtTokenizerOutputs* out, const float* d_pred, const float* gscal, vtStream stream) {
    int rc = vt_tokenizer_init_workspace(tk, ws, stream);
    if (!rc) rc = vt_tokenizer_pack(tk, params, ws, stream);
    if (!rc) rc = vt_tokenizer_encode(tk, params, video, ws, out, 1234u, stream);
    if (!rc) rc = vt_tokenizer_decode(tk, params, out->encoded, ws, out->pred_frames, stream);
    if (!rc) rc = vt_tokenizer_backward(tk, params, d_pred, gscal, ws, grads, 0, vt_tokenizer_num_backward_stages(tk), NULL, stream);
    return rc;
}

int main(void) {
    char msg[256];
    vtGemmNT g;
    vtTokenizerConfig c;
    vtTokenizer* tk = NULL;
    vtStackConfig sc = {2, 33, 128, 4, 2};
    vtStack* st = NULL;
    int rc;

    (void)example_training_step;
    printf("abi %d\n", vt_abi_version());
    memset(&g, 0, sizeof g);
    rc = vt_gemm_nt(&g, NULL);                      /* null operands: rejected before any launch */
    vt_last_error(msg, sizeof msg);
    printf("gemm_nt rc %d msg %s\n", rc, msg);
    memset(&c, 0, sizeof c);
    c.B = 1; c.C = 3; c.T = 4; c.S = 32; c.pt = 2; c.p = 16; c.D = 768; c.H = 12; c.depth_enc = 1; c.depth_dec = 1;
    c.Nq = 56; c.d = 24; c.K = 512; c.vq_mode = 0; c.l2_normalized = 1; c.inv_tau = 1.f; c.beta = 0.25f; c.codebook_w = 1.f;
    rc = vt_tokenizer_create(&c, &tk);              /* host-only planning: works without a GPU */
    printf("create rc %d stages %d ws %zu\n", rc, (int)vt_tokenizer_num_backward_stages(tk), vt_tokenizer_workspace_bytes(tk));
    vt_tokenizer_destroy(tk);
    c.H = 7;
    rc = vt_tokenizer_create(&c, &tk);
    vt_last_error(msg, sizeof msg);
    printf("bad create rc %d msg %s\n", rc, msg);
    rc = vt_stack_create(&sc, &st);
    printf("stack rc %d ws %zu\n", rc, vt_stack_workspace_bytes(st));
    vt_stack_destroy(st);
    return 0;
}
