#!/usr/bin/env python3
"""Dev tool: VGPR / SGPR / LDS / scratch of the kernels of the built libvidc.so whose name matches a regex (llvm-readelf notes)."""
import os, re, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vector_db_id_compression_amd import build
LLVM = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
work = tempfile.mkdtemp()
shutil.copy(build.LIB, os.path.join(work, "libvidc.so"))
subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "libvidc.so"], cwd=work, check=True, capture_output=True)
for o in sorted(p for p in os.listdir(work) if p.endswith("gfx950")):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", o], cwd=work, check=True, capture_output=True, text=True).stdout
    for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)"
                         r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
        lds, name, priv, sgpr, vgpr = m.groups()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        d = re.sub(r"\(.*", "", d.replace("void ", "").replace("(anonymous namespace)::", "").replace("vidc::dev::", ""))
        if pat.search(d):
            print(f"{d:60s} vgpr {vgpr:>4s} sgpr {sgpr:>4s} lds {lds:>6s} scratch {priv}")
shutil.rmtree(work)
