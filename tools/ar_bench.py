"""LARP_AR (the prior over the tokenizer's indices) on one MI355X: training step (fwd + loss + bwd) and KV-cache generation.
usage: python tools/ar_bench.py [size=B] [batch=8] [seq=1024] [gen_batch=16] [gen_tokens=256] [mode=both|train|gen]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import video_tokenizer_amd as vt  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "B"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
L = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
GB = int(sys.argv[4]) if len(sys.argv) > 4 else 16
GT = int(sys.argv[5]) if len(sys.argv) > 5 else 256
mode = sys.argv[6] if len(sys.argv) > 6 else "both"

torch.manual_seed(0)
m = vt.registry.make({"name": f"llama-abs-{size}", "args": dict(vocab_size=8192, max_seq_len=L, num_classes=101)}).cuda()
torch.nn.init.normal_(m.output.weight, std=0.02)
n_par = sum(p.numel() for p in m.parameters())
tok = torch.randint(0, 8192, (B, L), device="cuda")
lab = torch.randint(0, 101, (B,), device="cuda")
res = {"model": f"llama-abs-{size}", "params_M": round(n_par / 1e6, 1)}

if mode in ("both", "train"):
    m.train()

    def step():
        for p in m.parameters():
            p.grad = None
        _, loss = m(tok[:, :-1], lab, targets=tok)
        loss.backward()
        return loss

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    c = m.config
    hidden = m.layers[0].feed_forward.w1.weight.shape[0]
    per_tok = c.n_layer * (2 * c.dim * (4 * c.dim + 3 * hidden) + 2 * 2 * L * c.dim / 2) + 2 * c.dim * c.vocab_size     # fwd flops per token (causal attention = half)
    res["train"] = {"batch": B, "seq": L, "ms_per_step": round(dt * 1e3, 2), "tokens_per_s": round(B * L / dt), "model_tflops": round(3 * per_tok * B * L / dt / 1e12, 1),
                    "loss": round(loss.item(), 4)}

if mode in ("both", "gen"):
    m.eval()
    cond = torch.randint(0, 101, (GB,), device="cuda")
    for scale in (1.0, 2.0):
        torch.manual_seed(1)
        seq = m.sample(cond, cfg_scale=scale, temperature=1.0, top_k=0, top_p=1.0, seq_length=None if GT == L else None)
        m.reset_caches()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        seq = m.sample(cond, cfg_scale=scale, temperature=1.0, top_k=0, top_p=1.0)
        m.reset_caches()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[f"generate_cfg{scale:g}"] = {"batch": GB, "new_tokens": int(seq.shape[1]), "s": round(dt, 3), "tokens_per_s": round(GB * seq.shape[1] / dt),
                                         "ms_per_position": round(dt / seq.shape[1] * 1e3, 3)}
print(json.dumps(res))
