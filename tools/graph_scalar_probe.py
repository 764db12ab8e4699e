"""Root cause probe for the stale 0-dim read inside replayed graphs (found in round 3).

Hypothesis: a torch elementwise kernel reads a 0-dim operand through the SCALAR cache (s_load: the address is wave-uniform); the
producer wrote it with vector stores from another CU.  Between ordinary launches the runtime's kernel-start acquire invalidates the
scalar cache; between the nodes of a replayed hipGraph this build of the runtime does not (graph packet capture), so a line a previous
replay left in a CU's scalar cache can be served again.

Torch only, no library of this repo: graph = { a = mean|x| ; b = mean x^2 ; loss = a + 0.1 b ; prod = a * b ; seed = ones_like(loss) ;
g = seed * 3 } -- `seed` takes the pool block `a` frees, as the backward seed did in GraphedStep.  Replays on new x, a single-workgroup
kernel reading the graph's loss between replays (the round-3 trigger), values compared with eager.  Run in child processes under
several runtime settings; prints one line per setting.  usage: python tools/graph_scalar_probe.py"""
import os
import subprocess
import sys

CHILD = r'''
import torch
torch.manual_seed(0)
dev = "cuda"
N = 1 << 18
xs = [torch.randn(N, device=dev) * (1.0 + i) for i in range(12)]
x = xs[0].clone()
def body():
    a = x.abs().mean()
    b = (x * x).mean()
    loss = a + 0.1 * b
    prod = a * b
    del a
    seed = torch.ones_like(loss)
    g = seed * 3.0
    return loss, prod, g, b
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        body()
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    out = body()
bad_loss = bad_prod = bad_g = 0
first = None
for i, xi in enumerate(xs):
    x.copy_(xi)
    graph.replay()
    torch.cuda.synchronize()
    got = [o.item() for o in out]
    a = xi.abs().mean().item(); b = (xi * xi).mean().item()
    want = [a + 0.1 * b, a * b, 3.0, b]
    torch.equal(out[0], out[0].clone())          # a single-workgroup kernel reads the loss buffer (the round-3 trigger)
    torch.equal(out[2], out[2].clone())
    rel = [abs(g_ - w_) / max(abs(w_), 1e-9) for g_, w_ in zip(got, want)]
    bad_loss += rel[0] > 1e-4; bad_prod += rel[1] > 1e-4; bad_g += rel[2] > 1e-6
    if first is None and max(rel[:3]) > 1e-4:
        first = (i, got, want)
print(f"replays {len(xs)}: wrong loss {bad_loss}, wrong product {bad_prod}, wrong seed*3 {bad_g}; first mismatch {first}")
'''

settings = [("default", {}), ("DEBUG_CLR_GRAPH_PACKET_CAPTURE=0", {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}),
            ("HIP_FORCE_DEV_KERNARG=0", {"HIP_FORCE_DEV_KERNARG": "0"}), ("AMD_SERIALIZE_KERNEL=3", {"AMD_SERIALIZE_KERNEL": "3"})]
for name, env in settings:
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("replays")]
    print(f"{name:36s} {line[0] if line else 'FAILED: ' + r.stderr[-300:]}", flush=True)
