"""Build libfxhip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the
.so travels with the repo snapshot to the GPU box."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["fx_gemm.hip", "fx_gemm_bf16x3.hip", "fx_dw_adam_fwd.hip", "fx_norm_act.hip", "fx_fused_small.hip", "fx_heads.hip", "fx_block_bwd.hip", "fx_losses.hip", "fx_optim.hip", "fx_ingest.hip", "fx_gnn.hip"]
OUT = os.path.join(HERE, "libfxhip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = SOURCES + ["fx_common.h", "fx_reduce.h", "fx_small.h", "build.py"]
    return any(os.path.getmtime(os.path.join(HERE, s)) > t for s in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c",
               os.path.join(HERE, s), "-o", o]
        if verbose:
            print("[fxhip]", " ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode())
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[fxhip] compile failed: {s}\n{out.decode()}\n")
    if failed:
        raise RuntimeError("libfxhip build failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    if verbose:
        print("[fxhip]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
