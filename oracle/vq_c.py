"""ctypes wrapper over oracle/vq_oracle.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes

import numpy as np

from . import build as _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(_build.build())
        f32p = ctypes.POINTER(ctypes.c_float)
        i64p = ctypes.POINTER(ctypes.c_int64)
        _lib.vq_normalize_rows.argtypes = [f32p, ctypes.c_int64, ctypes.c_int, ctypes.c_int64, f32p, f32p, ctypes.c_float]
        _lib.vq_search.argtypes = [f32p, f32p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_float, i64p, f32p]
        _lib.vq_gather.argtypes = [f32p, f32p, i64p, ctypes.c_int64, ctypes.c_int, f32p, f32p, ctypes.POINTER(ctypes.c_double)]
        _lib.vq_codebook_grad.argtypes = [f32p, f32p, f32p, i64p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int,
                                          ctypes.c_int64, f32p]
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def normalize_rows(x, eps=1e-12):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    nrm = np.empty(x.shape[0], dtype=np.float32)
    lib().vq_normalize_rows(_p(x, ctypes.c_float), x.shape[0], x.shape[1], x.shape[1], _p(out, ctypes.c_float), _p(nrm, ctypes.c_float), eps)
    return out, nrm


def search(z, e, mode, inv_tau=1.0):
    """z (N,d), e (K,d) fp32, already normalised if the quantizer is l2_normalized.
    mode 'L' or 'D'. Returns (idx int64, score fp32)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    e = np.ascontiguousarray(e, dtype=np.float32)
    idx = np.empty(z.shape[0], dtype=np.int64)
    sc = np.empty(z.shape[0], dtype=np.float32)
    lib().vq_search(_p(z, ctypes.c_float), _p(e, ctypes.c_float), z.shape[0], e.shape[0], z.shape[1],
                    0 if mode == "L" else 1, np.float32(inv_tau), _p(idx, ctypes.c_int64), _p(sc, ctypes.c_float))
    return idx, sc


def gather(z, e, idx):
    z = np.ascontiguousarray(z, dtype=np.float32)
    e = np.ascontiguousarray(e, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    q = np.empty_like(z)
    rz = np.empty_like(z)
    tot = ctypes.c_double(0.0)
    lib().vq_gather(_p(z, ctypes.c_float), _p(e, ctypes.c_float), _p(idx, ctypes.c_int64), z.shape[0], z.shape[1],
                    _p(q, ctypes.c_float), _p(rz, ctypes.c_float), ctypes.byref(tot))
    return q, rz, tot.value


def vq_forward(z_in, emb_weight, mode, l2_normalized=True, temperature=0.03, beta=0.25, codebook_w=1.0):
    """Full fixed-order forward on (N,d) inputs: returns dict like the reference's (flat N)."""
    if l2_normalized:
        z, _ = normalize_rows(z_in)
        e, _ = normalize_rows(emb_weight)
    else:
        z = np.ascontiguousarray(z_in, np.float32)
        e = np.ascontiguousarray(emb_weight, np.float32)
    idx, sc = search(z, e, mode, 1.0 / temperature)
    q, rz, tot = gather(z, e, idx)
    mse = tot / z.size
    return {"z": z, "emb": e, "idx": idx, "score": sc, "q": q, "regularized_z": rz,
            "loss_commit": mse, "loss_codebook": mse, "loss_q": beta * mse + codebook_w * mse}


def codebook_grad(zn, e, wnorm, idx, s_b, normalize=True):
    """dW in the HIP path's summation order (vt_vq_backward): slabs of cb_slab_len(N) tokens, sequential inside."""
    zn = np.ascontiguousarray(zn, dtype=np.float32)
    e = np.ascontiguousarray(e, dtype=np.float32)
    wnorm = np.ascontiguousarray(wnorm, dtype=np.float32)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    n, d = zn.shape
    ns = min(32, max(1, (n + 511) // 512))
    slab = ((n + ns - 1) // ns + 63) // 64 * 64
    out = np.empty_like(e)
    lib().vq_codebook_grad(_p(zn, ctypes.c_float), _p(e, ctypes.c_float), _p(wnorm, ctypes.c_float), _p(idx, ctypes.c_int64), n, e.shape[0], d,
                           np.float32(s_b), int(normalize), slab, _p(out, ctypes.c_float))
    return out
