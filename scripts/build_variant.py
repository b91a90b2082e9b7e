"""Build an A/B variant of libfxhip.so: the named sources recompiled with extra -D flags, everything else linked from the
objects of the shipped build.  Output: build_tmp/libfxhip_<name>.so (select it with FXHIP_LIB=...).

    python scripts/build_variant.py prof fx_dw_adam_fwd.hip -DFT_PROFILE
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flexynesis_amd.csrc import build as b

name, rest = sys.argv[1], sys.argv[2:]
srcs = [a for a in rest if a.endswith(".hip")]
defs = [a for a in rest if not a.endswith(".hip")]
b.build(force=False, verbose=False)
out_dir = os.path.join(os.path.dirname(b.HERE), "..", "build_tmp")
out_dir = os.path.abspath(out_dir)
os.makedirs(out_dir, exist_ok=True)
objs = []
for s in b.SOURCES:
    o = os.path.join(b.HERE, s.replace(".hip", ".o"))
    if s in srcs:
        o = os.path.join(out_dir, f"{name}_{s.replace('.hip', '.o')}")
        cmd = ["/opt/rocm/bin/hipcc"] + b.FLAGS + defs + ["-c", os.path.join(b.HERE, s), "-o", o]
        if s == "fx_optim.hip":
            cmd.insert(1, f'-DFX_SOURCE_HASH="{b.source_hash()}"')
        subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs.append(o)
out = os.path.join(out_dir, f"libfxhip_{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
print(out)
