"""ctypes binding of libvt_hip.so (include/vt_hip.h).  PyTorch supplies device memory and the
stream; nothing here computes on the CPU.  Loading is lazy and LOUD: if the shared library is
missing (not built), `lib()` raises -- there is no fallback path."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VT_HIP_LIB") or os.path.join(_HERE, "libvt_hip.so")   # VT_HIP_LIB: A/B timing of two builds (tools/)
_lib = None

c_i32, c_i64, c_f32, c_u64, c_vp, c_sz = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t

EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_BF16_DGELU = 0, 1, 2, 3
TN_MAX_GROUP = 16


class RowMap(ctypes.Structure):
    _fields_ = [("grp", c_i32), ("stride", c_i64), ("off", c_i64)]


IDENT = RowMap(0, 0, 0)


class GemmNT(ctypes.Structure):
    _fields_ = [("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("epi", c_i32), ("out", c_vp), ("ldo", c_i64), ("out2", c_vp), ("ldo2", c_i64), ("bias", c_vp),
                ("residual", c_vp), ("ldr", c_i64), ("rowmod", c_vp), ("rowmod_period", c_i32), ("aux", c_vp),
                ("ldaux", c_i64), ("omap", RowMap), ("round_bf16", c_i32), ("colsum_partial", c_vp), ("out_scale", c_f32), ("tile", c_i32),
                ("splitk_ws", c_vp), ("splitk_ws_bytes", c_i64), ("splitk", c_i32)]


class GemmTN(ctypes.Structure):
    _fields_ = [("A", c_vp), ("lda", c_i64), ("B", c_vp), ("ldb", c_i64), ("M", c_i32), ("P", c_i32), ("Q", c_i32),
                ("out", c_vp), ("ldo", c_i64), ("p_lim", c_i32), ("q_lim", c_i32), ("row_perm", c_vp), ("tile", c_i32)]

# Tile generation the gemm_nt / gemm_tn_grouped wrappers below put into vtGemmNT.tile / vtGemmTN.tile when the caller
# passes none: 0 = the library's automatic choice.  Tests and tools set it to A/B the tile generations; it lives on the
# Python side only (the C library has no setting of its own, the field travels with every call).
GEMM_TILE = 0



# every symbol include/vt_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "vt_abi_version": (c_i32, []),
    "vt_last_error": (c_i32, [ctypes.c_char_p, c_sz]),
    "vt_gemm_nt": (c_i32, [ctypes.POINTER(GemmNT), c_vp]),
    "vt_gemm_nt_splitk_workspace_bytes": (c_sz, []),
    "vt_gemm_tn_grouped": (c_i32, [ctypes.POINTER(GemmTN), c_i32, c_vp]),
    "vt_layernorm_fwd": (c_i32, [c_vp, RowMap, c_vp, c_vp, c_f32, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "vt_layernorm_bwd_workspace_bytes": (c_sz, [c_i32]),
    "vt_layernorm_bwd": (c_i32, [c_vp, c_vp, RowMap, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vt_colsum_workspace_bytes": (c_sz, [c_i32]),
    "vt_colsum": (c_i32, [c_vp, c_i32, c_i64, RowMap, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vt_batch_sum": (c_i32, [c_vp, RowMap, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vt_cast_rows": (c_i32, [c_vp, RowMap, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "vt_zero_rows": (c_i32, [c_vp, c_vp, RowMap, c_i64, c_i32, c_vp]),
    "vt_sum_slabs": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "vt_assemble_rows": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "vt_pack_weight": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "vt_patchify": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vt_unpatchify": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp]),
    "vt_attention_fwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_attention_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_attention_causal_fwd": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_attention_causal_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_attention_fwd_rows": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_attention_bwd_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "vt_vq_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32]),
    "vt_vq_forward": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_u64, c_vp, c_vp,
                              c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vt_vq_forward_ctr": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_f32, c_f32, c_u64, c_vp, c_vp, c_vp,
                                  c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vt_vq_backward": (c_i32, [c_vp, c_i64, c_vp, c_f32, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                               c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "vt_vq_prep_codebook": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "vt_vq_gather": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "vt_adam_step": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp, c_f32, c_vp]),
    "vt_fsq_codebook_size": (c_i32, [ctypes.POINTER(c_i32), c_i32, ctypes.POINTER(c_i64)]),
    "vt_fsq_forward": (c_i32, [c_vp, c_i32, c_i64, c_i32, ctypes.POINTER(c_i32), c_vp, c_vp, c_vp]),
    "vt_fsq_backward": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, ctypes.POINTER(c_i32), c_vp, c_vp]),
    "vt_fsq_indices_to_codes": (c_i32, [c_vp, c_i64, c_i32, ctypes.POINTER(c_i32), c_vp, c_i32, c_vp]),
    "vt_qknorm_rope_fwd": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp]),
    "vt_qknorm_rope_bwd_workspace_bytes": (c_sz, []),
    "vt_qknorm_rope_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vt_sigmoid_gate_fwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "vt_sigmoid_gate_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vt_geglu_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "vt_geglu_bwd": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "vt_scale_rows": (c_i32, [c_vp, c_f32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vt_rmsnorm_fwd": (c_i32, [c_vp, c_vp, c_f32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "vt_rmsnorm_bwd_workspace_bytes": (c_sz, [c_i32]),
    "vt_rmsnorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "vt_swiglu_fwd": (c_i32, [c_vp, c_i64, c_i32, c_vp, c_vp]),
    "vt_swiglu_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp]),
    "vt_decode_attention": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "vt_decode_attention_step": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "vt_decode_norm_linear": (c_i32, [c_vp, c_vp, c_f32, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
}


class TokenizerConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("B", "C", "T", "S", "pt", "p", "D", "H", "depth_enc", "depth_dec", "Nq", "d", "K", "vq_mode", "l2_normalized")] + \
               [(n, c_f32) for n in ("inv_tau", "beta", "codebook_w")] + [("freeze_codebook", c_i32)]


BLOCK_FIELDS = ("norm1_w", "norm1_b", "qkv_w", "proj_w", "proj_b", "norm2_w", "norm2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")


class BlockTensors(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in BLOCK_FIELDS]


class PackJob(ctypes.Structure):
    _fields_ = [("w", c_vp), ("N", c_i32), ("K", c_i32), ("row_perm", c_vp), ("wb", c_vp), ("ldd", c_i64), ("wt", c_vp), ("lddT", c_i64)]


class StackConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("B", "L", "D", "H", "depth")]


class GatedStackConfig(ctypes.Structure):
    _fields_ = [(n, c_i32) for n in ("B", "L", "D", "H", "depth", "inner")]


GATED_FIELDS = ("to_qkv_w", "q_norm_w", "q_norm_b", "k_norm_w", "k_norm_b", "out_proj_w", "ln_w", "ln_b", "fc1_w", "fc2_w")


class GatedLayerTensors(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in GATED_FIELDS]


TENSOR_FIELDS = ("pe_w", "pe_b", "enc_patch_pe", "enc_query", "dec_latent_pe", "dec_patch_query", "dec_token_type", "in_w", "in_b",
                 "out_w", "out_b", "codebook", "head_norm_w", "head_norm_b", "head_w", "head_b")


class TokenizerTensors(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in TENSOR_FIELDS] + [("enc_blocks", ctypes.POINTER(BlockTensors)), ("dec_blocks", ctypes.POINTER(BlockTensors))]


OUTPUT_FIELDS = ("pred_frames", "encoded", "indices", "projected_z", "unregularized_z", "regularized_z", "emb", "losses", "input_norms")


class TokenizerOutputs(ctypes.Structure):
    _fields_ = [(n, c_vp) for n in OUTPUT_FIELDS]


_TT = ctypes.POINTER(TokenizerTensors)
ENGINE_SIGNATURES = {
    "vt_pack_weights_grouped": (c_i32, [ctypes.POINTER(PackJob), c_i32, c_vp]),
    "vt_stack_create": (c_i32, [ctypes.POINTER(StackConfig), ctypes.POINTER(c_vp)]),
    "vt_stack_destroy": (None, [c_vp]),
    "vt_stack_workspace_bytes": (c_sz, [c_vp]),
    "vt_stack_init_workspace": (c_i32, [c_vp, c_vp, c_vp]),
    "vt_stack_forward": (c_i32, [c_vp, ctypes.POINTER(BlockTensors), c_vp, c_vp, c_vp, c_vp]),
    "vt_stack_backward": (c_i32, [c_vp, ctypes.POINTER(BlockTensors), c_vp, c_vp, ctypes.POINTER(BlockTensors), c_vp, c_i32, c_vp]),
    "vt_gated_stack_create": (c_i32, [ctypes.POINTER(GatedStackConfig), ctypes.POINTER(c_vp)]),
    "vt_gated_stack_destroy": (None, [c_vp]),
    "vt_gated_stack_workspace_bytes": (c_sz, [c_vp]),
    "vt_gated_stack_init_workspace": (c_i32, [c_vp, c_vp, c_vp]),
    "vt_gated_stack_forward": (c_i32, [c_vp, ctypes.POINTER(GatedLayerTensors), c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "vt_gated_stack_backward": (c_i32, [c_vp, ctypes.POINTER(GatedLayerTensors), c_vp, c_vp, c_vp, c_vp, ctypes.POINTER(GatedLayerTensors), c_vp, c_vp]),
    "vt_tokenizer_create": (c_i32, [ctypes.POINTER(TokenizerConfig), ctypes.POINTER(c_vp)]),
    "vt_tokenizer_destroy": (None, [c_vp]),
    "vt_tokenizer_workspace_bytes": (c_sz, [c_vp]),
    "vt_tokenizer_init_workspace": (c_i32, [c_vp, c_vp, c_vp]),
    "vt_tokenizer_pack": (c_i32, [c_vp, _TT, c_vp, c_vp]),
    "vt_tokenizer_encode": (c_i32, [c_vp, _TT, c_vp, c_vp, ctypes.POINTER(TokenizerOutputs), c_u64, c_vp]),
    "vt_tokenizer_decode": (c_i32, [c_vp, _TT, c_vp, c_vp, c_vp, c_vp]),
    "vt_tokenizer_codes_to_encoded": (c_i32, [c_vp, _TT, c_vp, c_vp, c_vp, c_vp]),
    "vt_tokenizer_num_backward_stages": (c_i32, [c_vp]),
    "vt_tokenizer_set_seed_counter": (c_i32, [c_vp, c_vp]),
    "vt_tokenizer_set_split_k": (c_i32, [c_vp, c_i32]),
    "vt_tokenizer_set_wgrad_tail": (c_i32, [c_vp, c_i32]),
    "vt_tokenizer_set_wgrad_stream": (c_i32, [c_vp, c_vp]),
    "vt_tokenizer_set_wgrad_batch": (c_i32, [c_vp, c_i32]),
    "vt_stack_set_split_k": (c_i32, [c_vp, c_i32]),
    "vt_tokenizer_backward": (c_i32, [c_vp, _TT, c_vp, c_vp, c_vp, _TT, c_i32, c_i32, ctypes.POINTER(c_i32), c_vp]),
    "vt_tokenizer_set_data_parallel": (c_i32, [c_vp, c_i32]),
    "vt_tokenizer_backward_until_flush": (c_i32, [c_vp, _TT, c_vp, c_vp, c_vp, _TT, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i32), c_vp]),
}


def lib():
    """Load libvt_hip.so (built in-tree by video-tokenizer_amd/build.py).  Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: build it with `python video-tokenizer_amd/build.py` "
                               "(there is no CPU fallback for the tokenizer hot path)")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(ENGINE_SIGNATURES.items()):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib




class HipError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        buf = ctypes.create_string_buffer(512)
        lib().vt_last_error(buf, 512)
        raise HipError(f"{what} failed ({rc}): {buf.value.decode(errors='replace')}")


def ptr(t):
    """device pointer of a tensor (None -> NULL); refuses CPU tensors loudly."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("libvt_hip operates on GPU tensors only (no CPU path)")
    return ctypes.c_void_p(t.data_ptr())


def require_gpu(*tensors):
    """every compute entry point of this package is GPU-only: refuse CPU tensors before anything is allocated"""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipError("libvt_hip operates on GPU tensors only (no CPU path)")


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---------------------------------------------------------------------------------------------
# thin op-level wrappers (used by tests and by the module for the few ops outside the engine)
# ---------------------------------------------------------------------------------------------

GEMM_SPLITK = True       # False: gemm_nt never hands vt_gemm_nt a split-K workspace (A/B switch of tools and tests)
_SPLITK_WS = {}


def splitk_workspace(dev):
    """the zeroed split-K workspace of vt_gemm_nt for the current stream of `dev` (one per stream: the arrival counters in its
    first 4 KiB belong to one launch at a time)"""
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = torch.zeros(lib().vt_gemm_nt_splitk_workspace_bytes(), device=dev, dtype=torch.uint8)
        _SPLITK_WS[key] = ws
    return ws


def gemm_nt(A, B, epi=EPI_BF16, bias=None, out=None, out2=None, residual=None, rowmod=None, rowmod_period=0,
            aux=None, omap=None, round_bf16=False, out_rows=None, colsum_partial=None, tile=None, out_scale=0.0, splitk=1):
    """C = A @ B.T with the fused epilogue `epi`; A [M,K] bf16, B [N,K] bf16 (row-major, contiguous).
    splitk: 1 = never split K (default: the result's bits do not depend on how many rows share the launch), None = the library's
    automatic split for launches that leave most CUs idle (vtGemmNT.splitk_ws; a workspace per device and stream is kept here),
    2..8 = forced (tests)."""
    require_gpu(A, B)
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16
    M, K = A.shape
    N = B.shape[0]
    dev = A.device
    if out is None:
        rows = out_rows if out_rows is not None else M
        out = torch.empty(rows, N, device=dev, dtype=torch.float32 if epi == EPI_F32 else torch.bfloat16)
    if epi == EPI_BF16_GELU and out2 is None:
        out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    p = GemmNT()
    p.A, p.lda, p.B, p.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
    p.M, p.N, p.K, p.epi = M, N, K, epi
    p.out, p.ldo = out.data_ptr(), out.stride(0)
    p.out2, p.ldo2 = (out2.data_ptr(), out2.stride(0)) if out2 is not None else (None, 0)
    p.bias = bias.data_ptr() if bias is not None else None
    p.residual, p.ldr = (residual.data_ptr(), residual.stride(0)) if residual is not None else (None, 0)
    p.rowmod, p.rowmod_period = (rowmod.data_ptr(), rowmod_period) if rowmod is not None else (None, 0)
    p.aux, p.ldaux = (aux.data_ptr(), aux.stride(0)) if aux is not None else (None, 0)
    p.omap = omap if omap is not None else IDENT
    p.round_bf16 = int(round_bf16)
    p.colsum_partial = colsum_partial.data_ptr() if colsum_partial is not None else None
    p.tile = GEMM_TILE if tile is None else tile
    p.out_scale = float(out_scale)
    if splitk != 1 and GEMM_SPLITK:
        ws = splitk_workspace(dev)
        p.splitk_ws, p.splitk_ws_bytes, p.splitk = ws.data_ptr(), ws.numel(), 0 if splitk is None else int(splitk)
    else:
        p.splitk_ws, p.splitk_ws_bytes, p.splitk = None, 0, 1
    check(lib().vt_gemm_nt(ctypes.byref(p), stream()), "vt_gemm_nt")
    return (out, out2) if epi == EPI_BF16_GELU else out


def gemm_tn_grouped(problems):
    """problems: list of dicts(A=dY [M,P] bf16, B=X [M,Q] bf16, out fp32 [p_lim, >=q_lim], p_lim, q_lim, row_perm)."""
    arr = (GemmTN * len(problems))()
    for i, pr in enumerate(problems):
        A, B, out = pr["A"], pr["B"], pr["out"]
        g = arr[i]
        g.A, g.lda, g.B, g.ldb = A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0)
        g.M, g.P, g.Q = A.shape[0], A.shape[1], B.shape[1]
        g.out, g.ldo = out.data_ptr(), out.stride(0)
        g.p_lim = pr.get("p_lim", A.shape[1])
        g.q_lim = pr.get("q_lim", B.shape[1])
        rp = pr.get("row_perm")
        g.row_perm = rp.data_ptr() if rp is not None else None
        g.tile = pr.get("tile", GEMM_TILE)
    check(lib().vt_gemm_tn_grouped(arr, len(problems), stream()), "vt_gemm_tn_grouped")


def _ws(nbytes, dev):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)


def layernorm_fwd(x, gamma, beta, eps, rows=None, xmap=None, y=None):
    dim = x.shape[-1]
    rows = rows if rows is not None else x.numel() // dim
    y = torch.empty(rows, dim, device=x.device, dtype=torch.bfloat16) if y is None else y
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    check(lib().vt_layernorm_fwd(ptr(x), xmap or IDENT, ptr(gamma), ptr(beta), eps, rows, dim, ptr(y), ptr(mean), ptr(rstd), stream()), "vt_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, xmap=None, dx=None, dxb=None, want_dxsum=True):
    rows, dim = dy.shape
    dev = x.device
    dx = torch.empty_like(x) if dx is None else dx
    dxb = torch.empty(x.shape, device=dev, dtype=torch.bfloat16) if dxb is None else dxb
    dg, db = torch.empty(dim, device=dev), torch.empty(dim, device=dev)
    ds = torch.empty(dim, device=dev) if want_dxsum else None
    ws = _ws(lib().vt_layernorm_bwd_workspace_bytes(dim), dev)
    check(lib().vt_layernorm_bwd(ptr(dy), ptr(x), xmap or IDENT, ptr(gamma), ptr(mean), ptr(rstd), ptr(dres), rows, dim, ptr(dx),
                                 ptr(dxb), ptr(dg), ptr(db), ptr(ds), ptr(ws), stream()), "vt_layernorm_bwd")
    return dx, dxb, dg, db, ds


def colsum(src, rows=None, rmap=None):
    width = src.shape[-1]
    rows = rows if rows is not None else src.numel() // width
    out = torch.empty(width, device=src.device, dtype=torch.float32)
    ws = _ws(lib().vt_colsum_workspace_bytes(width), src.device)
    check(lib().vt_colsum(ptr(src), int(src.dtype == torch.bfloat16), src.stride(-2), rmap or IDENT, rows, width, ptr(out), ptr(ws), stream()), "vt_colsum")
    return out


def batch_sum(src, batch, n, rmap=None):
    dim = src.shape[-1]
    out = torch.empty(n, dim, device=src.device, dtype=torch.float32)
    check(lib().vt_batch_sum(ptr(src), rmap or IDENT, batch, n, dim, ptr(out), stream()), "vt_batch_sum")
    return out


def cast_rows(src, rows=None, rmap=None, ldd=None, dst=None):
    dim = src.shape[-1]
    rows = rows if rows is not None else src.numel() // dim
    ldd = ldd or (dst.stride(0) if dst is not None else dim)
    if dst is None:     # pad columns (ldd > dim) must read as zeros; a dense destination is fully overwritten
        dst = (torch.zeros if ldd != dim else torch.empty)(rows, ldd, device=src.device, dtype=torch.bfloat16)
    check(lib().vt_cast_rows(ptr(src), rmap or IDENT, rows, dim, ptr(dst), ldd, stream()), "vt_cast_rows")
    return dst


def assemble_rows(dst, seq, off, batch, n, src=None, table=None, vec=None):
    check(lib().vt_assemble_rows(ptr(dst), seq, off, batch, n, dst.shape[-1], ptr(src), ptr(table), ptr(vec), stream()), "vt_assemble_rows")


def pack_weight(w, row_perm=None, want_t=True, ldd=None, lddT=None, n_pad=None, k_pad=None):
    w2 = w.reshape(w.shape[0], -1)
    N, K = w2.shape
    ldd = ldd or (k_pad or K)
    lddT = lddT or (n_pad or N)
    dense = (n_pad or N) == N and (k_pad or K) == K and ldd == K and lddT == N      # no padding anywhere: fully overwritten
    alloc = torch.empty if dense else torch.zeros
    wb = alloc(n_pad or N, ldd, device=w.device, dtype=torch.bfloat16)
    wt = alloc(k_pad or K, lddT, device=w.device, dtype=torch.bfloat16) if want_t else None
    check(lib().vt_pack_weight(ptr(w2), N, K, ptr(row_perm), ptr(wb), ldd, ptr(wt), lddT, stream()), "vt_pack_weight")
    return wb, wt


def patchify(video, pt, p):
    B, C, T, S, _ = video.shape
    kp = C * pt * p * p
    nv = (T // pt) * (S // p) ** 2
    rows = torch.empty(B * nv, kp, device=video.device, dtype=torch.bfloat16)
    check(lib().vt_patchify(ptr(video), B, C, T, S, pt, p, ptr(rows), stream()), "vt_patchify")
    return rows


def unpatchify(rows, B, C, T, S, pt, p):
    video = torch.empty(B, C, T, S, S, device=rows.device, dtype=torch.float32)
    check(lib().vt_unpatchify(ptr(rows), B, C, T, S, pt, p, ptr(video), stream()), "vt_unpatchify")
    return video


def attention_fwd(qkv, B, L, H, hd=64, o=None, q_begin=0):
    """q_begin > 0: only the queries q_begin..L-1; o is then compact [B * (L - q_begin), H * hd]"""
    o = torch.empty(B * (L - q_begin), H * hd, device=qkv.device, dtype=torch.bfloat16) if o is None else o
    lse2 = torch.zeros(B, H, L, device=qkv.device, dtype=torch.float32) if q_begin else torch.empty(B, H, L, device=qkv.device, dtype=torch.float32)
    check(lib().vt_attention_fwd_rows(ptr(qkv), B, L, H, hd, q_begin, ptr(o), ptr(lse2), stream()), "vt_attention_fwd")
    return o, lse2


def attention_bwd(qkv, o, dO, lse2, B, L, H, hd=64, dqkv=None, q_begin=0):
    """the two-kernel backward (dQ + dK/dV, recompute from lse2); dqkv in the packed layout of qkv"""
    dqkv = torch.empty_like(qkv) if dqkv is None else dqkv
    delta = torch.zeros(B, H, L, device=qkv.device, dtype=torch.float32)
    check(lib().vt_attention_bwd_rows(ptr(qkv), ptr(o), ptr(dO), ptr(lse2), B, L, H, hd, q_begin, ptr(dqkv), ptr(delta), stream()), "vt_attention_bwd")
    return dqkv


def attention_causal_fwd(qkv, B, L, H):
    """F.scaled_dot_product_attention(q, k, v, is_causal=True) on the packed projection [B * L, 3 * H * 64]"""
    o = torch.empty(B * L, H * 64, device=qkv.device, dtype=torch.bfloat16)
    lse2 = torch.empty(B, H, L, device=qkv.device, dtype=torch.float32)
    check(lib().vt_attention_causal_fwd(ptr(qkv), B, L, H, ptr(o), ptr(lse2), stream()), "vt_attention_causal_fwd")
    return o, lse2


def attention_causal_bwd(qkv, o, dO, lse2, B, L, H):
    dqkv = torch.empty_like(qkv)
    delta = torch.zeros(B, H, L, device=qkv.device, dtype=torch.float32)
    check(lib().vt_attention_causal_bwd(ptr(qkv), ptr(o), ptr(dO), ptr(lse2), B, L, H, ptr(dqkv), ptr(delta), stream()), "vt_attention_causal_bwd")
    return dqkv


def rmsnorm_fwd(x, w, eps):
    require_gpu(x, w)
    rows, dim = x.shape
    y = torch.empty(rows, dim, device=x.device, dtype=torch.bfloat16)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    check(lib().vt_rmsnorm_fwd(ptr(x), ptr(w), eps, rows, dim, ptr(y), ptr(rstd), stream()), "vt_rmsnorm_fwd")
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres=None, want_bf16=False):
    require_gpu(dy, x, w, rstd)
    rows, dim = x.shape
    dx = torch.empty_like(x)
    dxb = torch.empty(rows, dim, device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    dw = torch.empty(dim, device=x.device, dtype=torch.float32)
    ws = _ws(lib().vt_rmsnorm_bwd_workspace_bytes(dim), x.device)
    check(lib().vt_rmsnorm_bwd(ptr(dy), ptr(x), ptr(w), ptr(rstd), ptr(dres), rows, dim, ptr(dx), ptr(dxb), ptr(dw), ptr(ws), stream()), "vt_rmsnorm_bwd")
    return dx, dxb, dw


def swiglu_fwd(h):
    require_gpu(h)
    M, I2 = h.shape
    a = torch.empty(M, I2 // 2, device=h.device, dtype=torch.bfloat16)
    check(lib().vt_swiglu_fwd(ptr(h), M, I2 // 2, ptr(a), stream()), "vt_swiglu_fwd")
    return a


def swiglu_bwd(da, h):
    require_gpu(da, h)
    M, I2 = h.shape
    dh = torch.empty_like(h)
    check(lib().vt_swiglu_bwd(ptr(da), ptr(h), M, I2 // 2, ptr(dh), stream()), "vt_swiglu_bwd")
    return dh


def decode_attention(q, k_cache, v_cache, n_keys):
    """q bf16 [B, H, 64]; caches bf16 [Bmax, H, Lmax, 64] (contiguous); keys 0..n_keys-1 visible -> o bf16 [B, H, 64]"""
    require_gpu(q, k_cache, v_cache)
    B, H, _ = q.shape
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape[1] == H and k_cache.shape[3] == 64 and k_cache.shape[0] >= B
    o = torch.empty_like(q)
    check(lib().vt_decode_attention(ptr(q), ptr(k_cache), ptr(v_cache), B, H, k_cache.shape[2], n_keys, ptr(o), stream()), "vt_decode_attention")
    return o


def decode_attention_step(qkv, k_cache, v_cache, pos_dev):
    """qkv bf16 [B, 3 * H * 64] = [q | k | v] of the new token; pos_dev int32 device tensor [1] = its position.  Stores k, v into the
    caches at that position and returns o bf16 [B, H * 64] over keys 0..pos.  No host synchronisation (graph-capturable)."""
    require_gpu(qkv, k_cache, v_cache, pos_dev)
    B = qkv.shape[0]
    H = k_cache.shape[1]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * H * 64 and k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape[0] >= B
    assert pos_dev.dtype == torch.int32 and pos_dev.numel() == 1
    o = torch.empty(B, H * 64, device=qkv.device, dtype=torch.bfloat16)
    check(lib().vt_decode_attention_step(ptr(qkv), ptr(k_cache), ptr(v_cache), B, H, k_cache.shape[2], ptr(pos_dev), ptr(o), stream()), "vt_decode_attention_step")
    return o


def decode_norm_linear(x, norm_w, eps, W, mode=0):
    """Linear(RMSNorm(x)) for M <= 64 rows in one launch.  x fp32 [M, K]; W bf16 [N, K]; mode 0 -> bf16 [M, N]; mode 1 (W = [w3 ; w1] in
    8 + 8-row slabs) -> SwiGLU output bf16 [M, N / 2]; mode 2 -> fp32 [M, N] of the bf16-rounded values."""
    require_gpu(x, norm_w, W)
    M, K = x.shape
    N = W.shape[0]
    if norm_w is None:
        assert x.dtype == torch.bfloat16           # already normalised operand: only the Linear and its epilogue are fused
    else:
        assert x.dtype == torch.float32 and norm_w.dtype == torch.float32
    assert W.dtype == torch.bfloat16 and x.is_contiguous() and W.is_contiguous() and W.shape[1] == K
    cols = N // 2 if mode == 1 else N
    out = torch.empty(M, cols, device=x.device, dtype=torch.float32 if mode == 2 else torch.bfloat16)
    check(lib().vt_decode_norm_linear(ptr(x), ptr(norm_w), eps, ptr(W), M, N, K, mode, ptr(out), cols, stream()), "vt_decode_norm_linear")
    return out


def vq_forward(z_in, codebook, mode, l2_normalized=True, inv_tau=1.0, beta=0.25, codebook_w=1.0, seed=0, ldp=0):
    """z_in fp32 [N, ldz] (first d columns used). Returns dict of saved tensors."""
    N = z_in.shape[0]
    K, d = codebook.shape
    dev = z_in.device
    o = {
        "E": torch.empty(K, d, device=dev), "wnorm": torch.empty(K, device=dev), "zn": torch.empty(N, d, device=dev),
        "znorm": torch.empty(N, device=dev), "idx": torch.empty(N, device=dev, dtype=torch.int64),
        "rz": torch.empty(N, d, device=dev), "losses": torch.empty(4, device=dev),
        "rz_pad": torch.zeros(N, ldp, device=dev, dtype=torch.bfloat16) if ldp else None,
    }
    ws = _ws(lib().vt_vq_workspace_bytes(N, K, d), dev)
    check(lib().vt_vq_forward(ptr(z_in), z_in.stride(0), ptr(codebook), N, K, d, mode, int(l2_normalized), inv_tau, beta, codebook_w,
                              seed, ptr(o["E"]), ptr(o["wnorm"]), ptr(o["zn"]), ptr(o["znorm"]), ptr(o["idx"]), ptr(o["rz"]),
                              ptr(o["rz_pad"]), ldp, ptr(o["losses"]), ptr(ws), stream()), "vt_vq_forward")
    return o


def vq_backward(g_rz, gscal, saved, beta=0.25, codebook_w=1.0, l2_normalized=True, ldp=0, need_dW=True):
    zn, E = saved["zn"], saved["E"]
    N, d = zn.shape
    K = E.shape[0]
    dev = zn.device
    dz = torch.empty(N, d, device=dev)
    dz_pad = torch.zeros(N, ldp, device=dev, dtype=torch.bfloat16) if ldp else None
    dW = torch.empty(K, d, device=dev) if need_dW else None          # frozen codebook: the dense K x d gradient is skipped
    ws = _ws(lib().vt_vq_workspace_bytes(N, K, d) if need_dW else 64, dev)
    check(lib().vt_vq_backward(ptr(g_rz), g_rz.stride(0) if g_rz is not None else 0, ptr(gscal), beta, codebook_w, ptr(zn),
                               ptr(saved["znorm"]), ptr(E), ptr(saved["wnorm"]), ptr(saved["idx"]), N, K, d, int(l2_normalized),
                               ptr(dz), ptr(dz_pad), ldp, ptr(dW), ptr(ws), stream()), "vt_vq_backward")
    return dz, dz_pad, dW


def _levels(levels):
    return (c_i32 * len(levels))(*[int(v) for v in levels])


def _fsq_dtype(t):
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"FSQ kernels take fp32 or bf16, got {t.dtype}")
    return int(t.dtype == torch.bfloat16)


def fsq_codebook_size(levels):
    """host-only (no GPU needed): prod(levels), with the library's own validation of the level list"""
    size = c_i64(0)
    check(lib().vt_fsq_codebook_size(_levels(levels), len(levels), ctypes.byref(size)), "vt_fsq_codebook_size")
    return size.value


def fsq_forward(z, levels, want_indices=True):
    """z [..., d] fp32/bf16 contiguous -> (codes like z, indices int32 [...])"""
    require_gpu(z)
    d = len(levels)
    assert z.is_contiguous() and z.shape[-1] == d
    N = z.numel() // d
    codes = torch.empty_like(z)
    idx = torch.empty(z.shape[:-1], device=z.device, dtype=torch.int32) if want_indices else None
    check(lib().vt_fsq_forward(ptr(z), _fsq_dtype(z), N, d, _levels(levels), ptr(codes), ptr(idx), stream()), "vt_fsq_forward")
    return codes, idx


def fsq_backward(z, dcodes, levels):
    require_gpu(z, dcodes)
    d = len(levels)
    assert z.is_contiguous() and dcodes.is_contiguous() and z.shape == dcodes.shape and z.dtype == dcodes.dtype
    dz = torch.empty_like(z)
    check(lib().vt_fsq_backward(ptr(z), ptr(dcodes), _fsq_dtype(z), z.numel() // d, d, _levels(levels), ptr(dz), stream()), "vt_fsq_backward")
    return dz


def fsq_indices_to_codes(indices, levels, dtype=torch.float32):
    require_gpu(indices)
    d = len(levels)
    idx = indices.to(torch.int32).contiguous()
    codes = torch.empty(*idx.shape, d, device=idx.device, dtype=dtype)
    check(lib().vt_fsq_indices_to_codes(ptr(idx), idx.numel(), d, _levels(levels), ptr(codes), _fsq_dtype(codes), stream()), "vt_fsq_indices_to_codes")
    return codes


# ---- glue of the TiTok-style block (csrc/vt_gated.hip) ----
def qknorm_rope_fwd(qkvg, L, H, q_w, q_b, k_w, k_b, eps, cos, sin):
    """qkvg bf16 [M, 4 * 64 H] -> packed bf16 [M, 3 * 64 H] (q, k normalised per head and rotated; v copied)"""
    require_gpu(qkvg, q_w, q_b, k_w, k_b, cos, sin)
    M = qkvg.shape[0]
    assert qkvg.dtype == torch.bfloat16 and qkvg.is_contiguous() and qkvg.shape[1] == 4 * 64 * H and cos.shape == (L, 32) == sin.shape
    out = torch.empty(M, 3 * 64 * H, device=qkvg.device, dtype=torch.bfloat16)
    check(lib().vt_qknorm_rope_fwd(ptr(qkvg), M, L, H, ptr(q_w), ptr(q_b), ptr(k_w), ptr(k_b), eps, ptr(cos), ptr(sin), ptr(out), stream()),
          "vt_qknorm_rope_fwd")
    return out


def qknorm_rope_bwd(qkvg, dqkv, L, H, q_w, k_w, eps, cos, sin, dqkvg):
    """writes columns 0..3D of dqkvg in place; returns (dq_w, dq_b, dk_w, dk_b)"""
    require_gpu(qkvg, dqkv, dqkvg)
    M = qkvg.shape[0]
    assert dqkv.is_contiguous() and dqkvg.is_contiguous() and dqkvg.shape == qkvg.shape and dqkv.shape == (M, 3 * 64 * H)
    grads = torch.empty(4, 64, device=qkvg.device)
    ws = _ws(lib().vt_qknorm_rope_bwd_workspace_bytes(), qkvg.device)
    check(lib().vt_qknorm_rope_bwd(ptr(qkvg), ptr(dqkv), M, L, H, ptr(q_w), ptr(k_w), eps, ptr(cos), ptr(sin), ptr(dqkvg), ptr(grads[0]),
                                   ptr(grads[1]), ptr(grads[2]), ptr(grads[3]), ptr(ws), stream()), "vt_qknorm_rope_bwd")
    return grads[0], grads[1], grads[2], grads[3]


def sigmoid_gate_fwd(o, qkvg):
    require_gpu(o, qkvg)
    M, D = o.shape
    assert o.is_contiguous() and qkvg.shape == (M, 4 * D)
    og = torch.empty_like(o)
    check(lib().vt_sigmoid_gate_fwd(ptr(o), ptr(qkvg), M, D, ptr(og), stream()), "vt_sigmoid_gate_fwd")
    return og


def sigmoid_gate_bwd(dog, o, qkvg, dqkvg):
    """returns d_o; writes columns 3D..4D of dqkvg in place"""
    require_gpu(dog, o, qkvg, dqkvg)
    M, D = o.shape
    assert dog.is_contiguous() and dog.shape == o.shape and dqkvg.shape == qkvg.shape
    d_o = torch.empty_like(o)
    check(lib().vt_sigmoid_gate_bwd(ptr(dog), ptr(o), ptr(qkvg), M, D, ptr(d_o), ptr(dqkvg), stream()), "vt_sigmoid_gate_bwd")
    return d_o


def geglu_fwd(h, lda=None):
    require_gpu(h)
    M, I2 = h.shape
    I = I2 // 2
    lda = I if lda is None else lda
    a = torch.empty(M, lda, device=h.device, dtype=torch.bfloat16) if lda == I else torch.zeros(M, lda, device=h.device, dtype=torch.bfloat16)
    check(lib().vt_geglu_fwd(ptr(h), M, I, ptr(a), lda, stream()), "vt_geglu_fwd")
    return a


def geglu_bwd(da, h):
    require_gpu(da, h)
    M, I2 = h.shape
    assert da.shape[0] == M and da.is_contiguous()
    dh = torch.empty_like(h)
    check(lib().vt_geglu_bwd(ptr(da), da.shape[1], ptr(h), M, I2 // 2, ptr(dh), stream()), "vt_geglu_bwd")
    return dh
