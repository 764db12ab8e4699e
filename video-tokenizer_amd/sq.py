"""Spherical ("sq") vector quantizer: nearest unit codeword by cosine over a frozen codebook.

Mirrors `VectorQuantizer` of /root/reference/models/model_new/quantizer/fsq.py:144-230 as the reference's
`LARPTokenizer(bottleneck_type='sq')` uses it (models/larp_tokenizer.py:225-229: n_embed=196_560, embed_dim=24,
l2_norm=True, beta=0.25, input_format='blc'): z <- z / |z|, E <- rows / |row|, idx = argmin(-z E^T) (first index on
ties), z_q = E[idx], loss = beta * mean_n(sum_d (sg(z_q) - z)^2) + mean_n(sum_d (z_q - sg(z))^2), output
z + sg(z_q - z).  The search is the tokenizer's exact-fp32 MFMA codebook search (vt_vq_forward, cosine mode): the
N x K score matrix (8192 x 196560 x 4 B = 6.4 GB in the reference) is never materialised.

The reference loads its codebook from a file on its author's disk (`leech_lattices_normalized.npy`, default argument
fsq.py:145); a checkpoint carries it as `bottleneck.embedding.weight`.  196 560 x 24 is the number of minimal vectors
of the Leech lattice, so when no file is given `leech_minimal_vectors()` generates exactly that set (unit-normalised;
the ROW ORDER is this build's own -- token ids are only comparable to the reference's through a checkpoint).
"""
import itertools
import os

import numpy as np
import torch
import torch.nn as nn

LEECH_K, LEECH_D = 196_560, 24


def golay_codewords():
    """The 4096 words of the extended binary Golay code, uint8 [4096, 24]: generator [I12 | B] with B the bordered circulant of
    (0 and the quadratic residues mod 11).  leech_minimal_vectors() checks the weight distribution (759 octads), which
    pins the code up to coordinate order."""
    qr = {(i * i) % 11 for i in range(1, 11)}
    first = np.array([1 if (j == 0 or j in qr) else 0 for j in range(11)], dtype=np.uint8)
    B = np.zeros((12, 12), dtype=np.uint8)
    for i in range(11):
        B[i, :11] = np.roll(first, -i)
        B[i, 11] = 1
    B[11, :11] = 1
    G = np.concatenate([np.eye(12, dtype=np.uint8), B], axis=1)
    msgs = ((np.arange(4096)[:, None] >> np.arange(12)[None, :]) & 1).astype(np.uint8)
    return (msgs @ G) % 2


def leech_minimal_vectors(normalized=True):
    """The 196 560 minimal vectors of the Leech lattice (squared norm 32 in the usual integer coordinates):
       1 104  of shape (+-4, +-4, 0^22);  97 152 of shape (+-2^8, 0^16) on the octads of the Golay code with an even number of
       minus signs;  98 304 of shape (-+3, +-1^23) with the signs of a Golay codeword.  float32 [196560, 24]."""
    C = golay_codewords()
    w = C.sum(1)
    assert sorted(np.unique(w).tolist()) == [0, 8, 12, 16, 24] and int((w == 8).sum()) == 759, "not the extended Golay code"
    out = []
    # shape 1
    v1 = []
    for i, j in itertools.combinations(range(24), 2):
        for si, sj in ((4, 4), (4, -4), (-4, 4), (-4, -4)):
            v = np.zeros(24, dtype=np.int8)
            v[i], v[j] = si, sj
            v1.append(v)
    out.append(np.stack(v1))
    # shape 2: sign patterns with an even number of minus signs over the 8 octad positions
    signs = np.array([s for s in itertools.product((1, -1), repeat=8) if s.count(-1) % 2 == 0], dtype=np.int8)   # [128, 8]
    oct_idx = np.stack([np.nonzero(c)[0] for c in C[w == 8]])                                                       # [759, 8]
    v2 = np.zeros((759, 128, 24), dtype=np.int8)
    for k in range(8):
        v2[np.arange(759)[:, None], np.arange(128)[None, :], oct_idx[:, k][:, None]] = 2 * signs[:, k][None, :]
    out.append(v2.reshape(-1, 24))
    # shape 3: eps = (-1)^c, v = eps with coordinate i replaced by -3 eps_i
    eps = (1 - 2 * C.astype(np.int8))                                                                               # [4096, 24]
    v3 = np.repeat(eps[:, None, :], 24, axis=1)
    ii = np.arange(24)
    v3[:, ii, ii] = -3 * eps[:, ii]
    out.append(v3.reshape(-1, 24))
    V = np.concatenate(out).astype(np.float32)
    assert V.shape == (LEECH_K, LEECH_D) and np.all((V * V).sum(1) == 32.0)
    return V / np.float32(np.sqrt(32.0)) if normalized else V


class VectorQuantizer(nn.Module):
    """fsq.py:144-230 (`input_format='blc'` only: that is how LARPTokenizer builds it)."""

    def __init__(self, n_embed, embed_dim, l2_norm, beta, input_format="bchw", predefined_codebook=None, freeze_codebook=True):
        super().__init__()
        if input_format != "blc":
            raise NotImplementedError("VectorQuantizer: only input_format='blc' is built (the tokenizer's use, larp_tokenizer.py:229)")
        if not l2_norm:
            raise NotImplementedError("VectorQuantizer: only l2_norm=True is built")
        if not freeze_codebook:
            raise NotImplementedError("VectorQuantizer: a trainable codebook is not built (the reference default freezes it, fsq.py:145,166)")
        self.n_embed, self.embed_dim, self.l2_norm, self.beta, self.input_format = n_embed, embed_dim, l2_norm, beta, input_format
        self.embedding = nn.Embedding(n_embed, embed_dim)
        self.embedding.weight.data.uniform_(-1 / n_embed, 1 / n_embed)
        self.bits_per_index = int(np.ceil(np.log2(n_embed)))
        if predefined_codebook is not None:
            if isinstance(predefined_codebook, str):
                if os.path.exists(predefined_codebook):
                    cb = np.load(predefined_codebook, allow_pickle=False)
                elif (n_embed, embed_dim) == (LEECH_K, LEECH_D):
                    cb = leech_minimal_vectors()        # the file of the reference's default path is not shipped: generate the set
                else:
                    raise FileNotFoundError(predefined_codebook)
            else:
                cb = np.asarray(predefined_codebook)
            assert cb.shape == (n_embed, embed_dim), "Predefined codebook has incorrect shape"
            self.embedding.weight.data.copy_(torch.from_numpy(np.ascontiguousarray(cb, dtype=np.float32)))
        self.embedding.weight.requires_grad = False

    def forward(self, z):
        """returns {'output', 'loss_codebook'} like fsq.py:170-207; `indices` is attached as an extra (detached) key"""
        from .functional import VectorQuantize
        assert z.dim() == 3 and z.shape[-1] == self.embed_dim
        rz, idx, lq, lc, lcb, zn, emb = VectorQuantize.apply(z.float(), self.embedding.weight, 1, True, 1.0, float(self.beta), 1.0, 0)
        # engine loss_q = beta * mean_{n,d} (q - z)^2 + mean_{n,d} (q - z)^2; the reference sums over d and averages over tokens
        return {"output": rz, "loss_codebook": lq * float(self.embed_dim), "indices": idx.reshape(z.shape[0], z.shape[1])}

    def get_entropy(self, count, eps=1e-4):
        probs = (count + eps) / (count + eps).sum()
        return -(probs * torch.log(probs)).sum()

    def get_codebook_entry(self, indices):
        from .functional import codebook_entries
        return codebook_entries(indices, self.embedding.weight, True)
