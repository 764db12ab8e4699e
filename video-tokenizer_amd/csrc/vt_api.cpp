// C-ABI plumbing shared by every entry point of libvt_hip.so: version and thread-local error text.
#include <stdarg.h>

#include "vt_common.h"

static thread_local char g_err[512] = "";

void vt_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int vt_abi_version(void) { return 8; }  // 8: vt_tokenizer_set_data_parallel (no longer implied by the weight-gradient tail), vt_tokenizer_backward_until_flush; 7: vt_tokenizer_set_wgrad_tail, vt_tokenizer_set_wgrad_stream, vt_tokenizer_set_wgrad_batch; 6: vt_attention_bwd_fused* and vt_tokenizer_status_offset removed (the five-product backward lost to the two-kernel form), vtGemmTN.tile 7; 5: vtGemmNT.splitk_ws / splitk_ws_bytes / splitk, vt_gemm_nt_splitk_workspace_bytes; 4: vt_attention_bwd_fused*, vt_vq_forward_ctr, vt_tokenizer_set_seed_counter / _status_offset; 2: vt_vq_backward takes a workspace, vtGemmNT.colsum_partial; 3: vtGemmNT/vtGemmTN.tile per call, vt_set_gemm_variant removed, vtTokenizerConfig.freeze_codebook, vt_vq_backward dW optional

extern "C" int vt_last_error(char* buf, size_t n) {
    if (!buf || n == 0) return VT_ERR_INVALID;
    strncpy(buf, g_err, n - 1);
    buf[n - 1] = 0;
    return VT_OK;
}
