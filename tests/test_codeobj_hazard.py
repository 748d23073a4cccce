"""CPU: a static gate on the built gfx950 code objects for the register-index hazard of DESIGN section 11.

What the evidence supports (profiles/r04c_pair_forms.txt): on gfx950 a VOP3-encoded instruction that reads or writes a register
through the index must not follow `s_set_gpr_idx_on`; the shipped kernels only ever put the compiler's own pattern -- a VOP1
`v_mov_b32` (optionally behind `s_nop`) -- between `s_set_gpr_idx_on` and `s_set_gpr_idx_off`.  That was a convention kept by
reading the sources; this test turns it into a gate on what the assembler actually emitted: every index region of every kernel
in libvidc.so is disassembled with llvm-objdump and must consist of `s_nop` and VOP1-encoded `v_mov_b32` only, and must be
closed before any branch, label or end of program."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _disassemble(tmp_path):
    from vector_db_id_compression_amd import build

    lib = build.build()
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(lib, work / "libvidc.so")  # (the bundles are written next to the file that is read)
    subprocess.run([OBJDUMP, "--offloading", "libvidc.so"], cwd=work, check=True, capture_output=True)
    objs = sorted(p for p in os.listdir(work) if p.endswith("gfx950"))
    assert objs, "no gfx950 code object in libvidc.so"
    text = []
    for o in objs:
        r = subprocess.run([OBJDUMP, "-d", o], cwd=work, check=True, capture_output=True, text=True)
        text.append((o, r.stdout.split("\n")))
    return text


def index_regions(lines):
    """[(line number of s_set_gpr_idx_on, [(mnemonic, first encoding word, text)], closed)] of one disassembly."""
    out = []
    i = 0
    while i < len(lines):
        if "s_set_gpr_idx_on" in lines[i]:
            body, closed, j = [], False, i + 1
            while j < len(lines):
                t = lines[j].strip()
                j += 1
                if not t:
                    continue
                if "s_set_gpr_idx_off" in t:
                    closed = True
                    break
                m = re.match(r"(\S+)\s.*//\s*[0-9A-Fa-f]+:\s*([0-9A-Fa-f]{8})", t)
                if m is None:  # a label, a symbol line, an end of section: the region ran away
                    body.append(("<not an instruction>", 0, t))
                    break
                body.append((m.group(1), int(m.group(2), 16), t))
                if len(body) > 8:
                    break
            out.append((i + 1, body, closed))
            i = j
        else:
            i += 1
    return out


def test_index_regions_hold_only_the_vop1_move_pattern(tmp_path):
    total = 0
    bad = []
    for name, lines in _disassemble(tmp_path):
        for lineno, body, closed in index_regions(lines):
            total += 1
            if not closed:
                bad.append((name, lineno, "region not closed by s_set_gpr_idx_off", body))
                continue
            moves = 0
            for mnem, word, text in body:
                if mnem == "s_nop":
                    continue
                # VOP1: bits 31..25 = 0111111; llvm prints the _e32 suffix for it (a VOP3 form would be _e64 / 0xD1......)
                if mnem == "v_mov_b32_e32" and (word >> 25) == 0x3F:
                    moves += 1
                    continue
                bad.append((name, lineno, f"'{mnem}' inside an s_set_gpr_idx_on region", text))
            if moves == 0 and not any(b[1] == lineno and b[0] == name for b in bad):
                bad.append((name, lineno, "region without an indexed move", body))
    assert total >= 50, f"only {total} register-index regions found: the disassembly did not see the ROC kernels"
    assert not bad, "\n".join(map(str, bad[:10]))


def test_the_checker_flags_the_round3_form():
    """The construct that corrupted other wavefronts (form 1 of profiles/r04c_pair_forms.txt) and a runaway region, as text."""
    lines = [
        "\ts_set_gpr_idx_on s39, gpr_idx(SRC0,DST)                   // 000000020258: BF110927",
        "\tv_cndmask_b32_e64 v64, v64, v3, s[10:11]                   // 00000002025C: D1000040 002A0740",
        "\ts_set_gpr_idx_off                                          // 000000020264: BF9C0000",
        "\ts_set_gpr_idx_on s39, gpr_idx(DST)                         // 000000020374: BF110827",
        "\tv_mov_b32_e32 v2, v43                                      // 000000020378: 7E04032B",
        "\ts_cbranch_scc1 65530                                       // 00000002037C: BF85FFFA",
        "\ts_set_gpr_idx_off                                          // 000000020380: BF9C0000",
        "\ts_set_gpr_idx_on s39, gpr_idx(DST)                         // 000000020384: BF110827",
        "\ts_nop 0                                                    // 000000020388: BF800000",
        "\tv_mov_b32_e32 v2, v43                                      // 00000002038C: 7E04032B",
        "\ts_set_gpr_idx_off                                          // 000000020390: BF9C0000",
    ]
    regs = index_regions(lines)
    assert len(regs) == 3
    assert regs[0][1][0][0] == "v_cndmask_b32_e64" and (regs[0][1][0][1] >> 25) != 0x3F
    assert [m for m, _, _ in regs[1][1]] == ["v_mov_b32_e32", "s_cbranch_scc1"]
    assert [m for m, _, _ in regs[2][1]] == ["s_nop", "v_mov_b32_e32"] and regs[2][2]
