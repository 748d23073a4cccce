"""Static gate on the resources of the built gfx950 kernels (CPU test: reads the code objects of libvidc.so with llvm-readelf).

DESIGN sections 12 / 13 reason about what runs next to what on a SIMD from the kernels' register and LDS footprints (512 VGPRs and
eight wavefront slots per SIMD, 160 KiB of LDS per CU).  Those numbers are properties of the BUILD, so they are checked on the build:

* no kernel uses scratch memory (a register spill in a chain loop or a streaming kernel is a silent 2x; the one deliberate
  experiment with spills -- the chain kernels held to 128 VGPRs -- made S2 6 ms slower and is not in the tree);
* the hand-scheduled chain kernels stay at two wavefronts per SIMD or better (<= 256 VGPRs; they are 192 .. 213 today), the
  register decoder for 65 .. 1024-id lists is 256 by design (192 id slots pinned to v64 .. v255);
* the bandwidth-bound kernels (Elias-Fano, packed bits, the stream compaction, the gather) keep at least five wavefronts per SIMD
  (<= 96 VGPRs) and a few KiB of LDS, so that their occupancy hides HBM latency;
* the lane-per-list ROC kernels stay within the LDS strips the launch code assumes (comments at the launches in csrc/roc.hip).
"""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernels(tmp_path):
    from vector_db_id_compression_amd import build

    lib = build.build()
    if not (os.path.exists(f"{LLVM}/llvm-objdump") and os.path.exists(f"{LLVM}/llvm-readelf")):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not found")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "libvidc.so")
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "libvidc.so"], cwd=work, check=True, capture_output=True)
    out = {}
    for o in sorted(p for p in os.listdir(work) if p.endswith("gfx950")):
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", o], cwd=work, check=True, capture_output=True, text=True).stdout
        for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)"
                             r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", notes, re.S):
            lds, name, priv, sgpr, vgpr = m.groups()
            out[name] = {"lds": int(lds), "scratch": int(priv), "sgpr": int(sgpr), "vgpr": int(vgpr)}
    assert len(out) > 100, f"only {len(out)} kernels found in the gfx950 code objects"
    names = list(out)
    cf = shutil.which("c++filt")
    dem = subprocess.run([cf] + names, capture_output=True, text=True).stdout.split("\n") if cf else names
    res = {}
    for n, d in zip(names, dem):
        short = re.sub(r"\(.*", "", d).replace("void ", "").replace("(anonymous namespace)::", "").replace("vidc::dev::", "")
        res[short or n] = out[n]
    return res


def test_no_kernel_spills_and_the_footprints_design_reasons_with(tmp_path):
    ks = _kernels(tmp_path)
    spilled = {k: v["scratch"] for k, v in ks.items() if v["scratch"]}
    assert not spilled, f"kernels with scratch memory (register spills): {spilled}"

    def group(pat):
        g = {k: v for k, v in ks.items() if re.search(pat, k)}
        assert g, f"no kernel matches {pat}: renamed?"
        return g

    # hand-scheduled chain kernels: two wavefronts per SIMD at least
    for k, v in group(r"^k_roc_(encode_u2|encode_r2|decode_u2|decode_b2)<").items():
        assert v["vgpr"] <= 256, (k, v)
    # ids in registers: 192 slots pinned to v64 .. v255
    for k, v in group(r"^k_roc_decode_lane_reg<").items():
        assert v["vgpr"] == 256 and v["lds"] <= 30 * 1024, (k, v)
    # bucket-row lane decoders / lane encoders: the strips the launches count on (12.8 / 17.9 / 27.1 KiB; 14.6 / 22 / 32 / 51 KiB)
    for k, v in group(r"^k_roc_decode_lane<").items():
        assert v["vgpr"] <= 128 and v["lds"] <= 28 * 1024, (k, v)
    for k, v in group(r"^k_roc_encode_lane<").items():
        assert v["vgpr"] <= 64 and v["lds"] <= 52 * 1024, (k, v)
    # row-per-list kernels: four lists per wavefront, LDS is dynamic
    for k, v in group(r"^k_roc_(encode|decode)_grp<").items():
        assert v["vgpr"] <= 64, (k, v)
    # bandwidth-bound kernels: five wavefronts per SIMD or more, small static LDS
    for k, v in group(r"^(k_ef_lowhigh32<|k_ef_decode_rec<|k_packed_encode<|k_packed_decode<|k_roc_compact|k_gather_ids)").items():
        assert v["vgpr"] <= 96 and v["lds"] <= 8 * 1024, (k, v)
