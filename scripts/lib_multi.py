"""run `python <script> <args>` with each alternative build pea_diffusion_amd/libpea_hip_<tag>.so in turn (timing probes)
usage: python scripts/lib_multi.py tag1,tag2 <script> [args...]"""
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(root, "pea_diffusion_amd", "libpea_hip.so")
base = "/tmp/libpea_hip_base.so"
shutil.copy(lib, base)
try:
    for tag in ["base"] + sys.argv[1].split(","):
        src = base if tag == "base" else os.path.join(root, "pea_diffusion_amd", f"libpea_hip_{tag}.so")
        shutil.copy(src, lib)
        out = subprocess.run([sys.executable] + sys.argv[2:], capture_output=True, text=True, cwd=root)
        print(f"===== {tag}")
        print("\n".join(l for l in (out.stdout + out.stderr).splitlines() if "amdgpu.ids" not in l), flush=True)
finally:
    shutil.copy(base, lib)
