"""A/B of two builds of libpea_hip.so on one box: runs `python <script> <args>` alternately with the in-tree library and
with pea_diffusion_amd/libpea_hip_alt.so (same sources, different compile flags) and prints both outputs.
usage: python scripts/lib_ab.py <rounds> <script> [args...]"""
import os, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "pea_diffusion_amd", "libpea_hip.so")
alt = os.path.join(root, "pea_diffusion_amd", "libpea_hip_alt.so")
base = "/tmp/libpea_hip_base.so"
shutil.copy(lib, base)
try:
    for r in range(int(sys.argv[1])):
        for name, src in (("base", base), ("alt", alt)):
            shutil.copy(src, lib)
            out = subprocess.run([sys.executable] + sys.argv[2:], capture_output=True, text=True, cwd=root)
            print(f"===== round {r} {name}")
            print("\n".join(l for l in (out.stdout + out.stderr).splitlines() if "amdgpu.ids" not in l), flush=True)
finally:
    shutil.copy(base, lib)
