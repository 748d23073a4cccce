#!/usr/bin/env python3
"""Dev tool: per-wavefront view of a tools/pmc_cmd.sh result (gpurun_out/<tag>/pmc_<name>.json)."""
import json, sys
d = json.load(open(sys.argv[1]))["per_kernel_per_dispatch"]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in sorted(d.items()):
    if pat not in k or "SQ_WAVES" not in c:
        continue
    w = c["SQ_WAVES"]
    f = lambda n: c.get(n, 0) / w
    print(f"{k}: {c.get('GRBM_GUI_ACTIVE', 0) / 8 / 2400:.1f} us  waves {w:.0f}\n"
          f"   per wave: VALU {f('SQ_INSTS_VALU'):.0f} SALU {f('SQ_INSTS_SALU'):.0f} LDS {f('SQ_INSTS_LDS'):.0f} VMEM rd {f('SQ_INSTS_VMEM_RD'):.0f} "
          f"wr {f('SQ_INSTS_VMEM_WR'):.0f} branch {f('SQ_INSTS_BRANCH'):.0f}\n"
          f"   cycles per wave: resident {4 * f('SQ_WAVE_CYCLES'):.0f} issuing {4 * f('SQ_ACTIVE_INST_ANY'):.0f} wait_any {4 * f('SQ_WAIT_ANY'):.0f} "
          f"wait_inst {4 * f('SQ_WAIT_INST_ANY'):.0f} wait_lds {4 * f('SQ_WAIT_INST_LDS'):.0f}\n"
          f"   LDS: conflict cycles {f('SQ_LDS_BANK_CONFLICT'):.0f} of {f('SQ_LDS_IDX_ACTIVE'):.0f}; HBM: fetch {c.get('FETCH_SIZE', 0) / 1e3:.1f} MB(*) "
          f"write {c.get('WRITE_SIZE', 0) / 1e3:.1f} MB; TA busy {c.get('TA_BUSY_avr', 0) / max(c.get('GRBM_GUI_ACTIVE', 1) / 8, 1):.2f}")
