"""Build libfxhip.so for gfx950 with hipcc (cross-compiles without a GPU).  In-tree output so the
.so travels with the repo snapshot to the GPU box.

The library carries the SHA-256 of the sources it was built from (``fx_source_hash()``; the literal sits in the binary
behind the marker ``FXSRCHASH:``).  ``needs_build`` compares that with the hash of the sources as they are now, so a
stale binary next to edited sources is rebuilt whatever the file times say."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["fx_gemm.hip", "fx_gemm_bf16x3.hip", "fx_dw_adam_fwd.hip", "fx_norm_act.hip", "fx_fused_small.hip", "fx_heads.hip", "fx_small_linear.hip", "fx_enc_tail.hip", "fx_assembly.hip",
           "fx_block_bwd.hip", "fx_losses.hip", "fx_optim.hip", "fx_ingest.hip", "fx_gnn.hip", "fx_sampling.hip", "fx_runtime.hip"]
HEADERS = ["fx_common.h", "fx_reduce.h", "fx_small.h", "fx_loss_dev.h", "fx_chain_prof.h"]
# packed fp32 VALU ops are off: a v_pk_fma_f32 whose low source register was written by the preceding VALU instructions drops
# that term in lanes 48..63 when another wave on the SIMD streams ds_read_b128 results into back-to-back MFMAs (measured:
# scripts/pkfma_hazard_probe.hip) -- i.e. whenever a small fp32 kernel shares a CU with one of the
# MFMA kernels, which the step's side branches and trials in flight make routine.  (hipcc hands -Xclang options to the host
# pass too, which prints "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)" once per file:
# harmless; -Xarch_device does not accept -Xclang.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
OUT = os.path.join(HERE, "libfxhip.so")
MARK = b"FXSRCHASH:"


def source_hash() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update(name.encode())
        h.update(open(os.path.join(HERE, name), "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def built_hash():
    """The hash baked into the existing library (read from the file: no HIP runtime needed), or None."""
    if not os.path.exists(OUT):
        return None
    blob = open(OUT, "rb").read()
    i = blob.find(MARK)
    if i < 0:
        return None
    return blob[i + len(MARK): i + len(MARK) + 64].decode("ascii", "replace")


def needs_build():
    return built_hash() != source_hash()


def build(force=False, verbose=True):
    want = source_hash()
    if not force and built_hash() == want:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, s.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(HERE, s), "-o", o]
        if s == "fx_optim.hip":
            cmd.insert(1, f'-DFX_SOURCE_HASH="{want}"')
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
    if built_hash() != want:
        raise RuntimeError("libfxhip build: the source hash is missing from the binary")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
