"""Where the time of LARP_AR generation goes on the host: total, graph capture, host time inside CUDAGraph.replay().  python tools/gen_breakdown.py (GPU)"""
import sys, time, torch
sys.path.insert(0, ".")
import video_tokenizer_amd as vt
from video_tokenizer_amd import larp_ar as A
m = vt.registry.make({"name": "llama-abs-L", "args": dict(vocab_size=8192, max_seq_len=1024, num_classes=101)}).cuda().eval()
torch.nn.init.normal_(m.output.weight, std=0.02)
cond = torch.randint(0, 101, (16,), device="cuda")
for rep in range(2):
    for n_new in (64, 1024):
        m.config.max_seq_len = m.max_seq_length = 1024
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with m.sampling():
            out = A.generate(m, cond, 1024, cfg_scale=1.0, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True) if n_new == 1024 else None
        m.reset_caches()
        torch.cuda.synchronize()
        if out is not None:
            print("full generate 1024:", round(time.perf_counter() - t0, 3), "s")
# replay-only timing: patch CUDAGraph.replay to count
import torch.cuda
orig = torch.cuda.CUDAGraph.replay
cnt = [0, 0.0]
def rep(self):
    cnt[0] += 1
    t = time.perf_counter()
    r = orig(self)
    cnt[1] += time.perf_counter() - t
    return r
torch.cuda.CUDAGraph.replay = rep
orig_cap = torch.cuda.graph.__exit__
tcap = [0.0]
t_enter = [0.0]
oe = torch.cuda.graph.__enter__
def en(self):
    torch.cuda.synchronize(); t_enter[0] = time.perf_counter()
    return oe(self)
def ex(self, *a):
    r = orig_cap(self, *a)
    torch.cuda.synchronize(); tcap[0] += time.perf_counter() - t_enter[0]
    return r
torch.cuda.graph.__enter__ = en
torch.cuda.graph.__exit__ = ex
torch.cuda.synchronize(); t0 = time.perf_counter()
with m.sampling():
    out = A.generate(m, cond, 1024, cfg_scale=1.0, temperature=1.0, top_k=0, top_p=1.0, sample_logits=True)
torch.cuda.synchronize()
print("total", round(time.perf_counter() - t0, 3), "capture", round(tcap[0], 3), "replays", cnt[0], "host time inside replay()", round(cnt[1], 3))
