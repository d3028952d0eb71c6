"""Build a variant of the library for A/B timing on one box: one translation unit compiled with extra flags, linked with the
standard objects of the other units.
    python scripts/variant.py NAME rast_render_bwd.hip -DSOAR_X=1 ...   ->  soar_amd/_lib/variants/NAME.so
Run it on the GPU box with  python scripts/ab_lib.py soar_amd/_lib/variants/NAME.so [bench.py arguments]."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from soar_amd import build as B

name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
out_dir = os.path.join(B.OUT_DIR, "variants")
os.makedirs(out_dir, exist_ok=True)
obj = os.path.join(out_dir, name + ".o")
cmd = [B.HIPCC] + B.COMMON_FLAGS + B.EXTRA_FLAGS.get(unit, []) + flags + ["-c", os.path.join(B.CSRC, unit), "-o", obj]
subprocess.check_call(cmd)
objs = [obj if s == unit else os.path.join(B.OBJ_DIR, s.replace(".hip", ".o")) for s in B.SOURCES]
lib = os.path.join(out_dir, name + ".so")
subprocess.check_call([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"])
os.remove(obj)
print(lib)
