"""Build libvt_hip.so (all gfx950 kernels + the C ABI of include/vt_hip.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels to the
GPU box with the snapshot.  Usage: python video-tokenizer_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvt_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["vt_api.cpp", "vt_gemm.hip", "vt_gemm192.hip", "vt_norm.hip", "vt_patch.hip", "vt_vq.hip", "vt_attention.hip", "vt_optim.hip", "vt_fsq.hip", "vt_gated.hip", "vt_ar.hip", "vt_engine.hip", "vt_gated_engine.hip"]
AUDIT_NO_SPILL = {"vt_gemm.hip", "vt_gemm192.hip", "vt_attention.hip"}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "vt_hip.h")]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        op = os.path.join(HERE, "_obj", os.path.splitext(src)[0] + ".o")
        objs.append(op)
        if force or _newer(sp, op) or any(_newer(d, op) for d in deps):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", sp, "-o", op]
            if src in AUDIT_NO_SPILL:
                cmd.append("-Rpass-analysis=kernel-resource-usage")
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        text = out.decode()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- {src} failed ---\n{text}\n")
            continue
        if src in AUDIT_NO_SPILL:
            # these kernels load LDS fragments / issue LDS-DMA through inline asm; a register spill would move an
            # asm destination before its data has landed (silent wrong results), so spills are a build error
            # (SGPR spills go to VGPR lanes by v_writelane, no memory involved: allowed)
            bad = [ln for ln in text.splitlines() if ("VGPRs Spill:" in ln or "ScratchSize" in ln)
                   and not ln.rstrip().endswith(" 0 [-Rpass-analysis=kernel-resource-usage]")]
            if bad:
                failed = True
                os.remove(os.path.join(HERE, "_obj", os.path.splitext(src)[0] + ".o"))  # so the next build re-checks
                sys.stderr.write(f"--- {src}: VGPR spills / scratch in an inline-asm kernel ---\n" + "\n".join(bad[:8]) + "\n")
        elif verbose and text:
            sys.stderr.write(text)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
