#!/usr/bin/env python3
"""Per-kernel matrix-pipe and vector-ALU occupancy from one rocprofv3 --pmc pass over the bench
(counters: SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE).

    python tools/pmc_mfma_table.py gpurun_out/<pmc dir> > profiles/rNN_x_pmc_mfma.md

Units (MI355X_MICROARCH.md, cycle-constants table): SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs
(32 per 32x32x16 bf16 MFMA, 16 per 16x16x32); SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES count quad-cycles; GRBM_GUI_ACTIVE is summed
over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 is the kernel's length in shader cycles.
    mfma pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)
i.e. the share of the kernel's cycles in which a SIMD's matrix pipe executes, chip average: x 2.5 PFLOP/s x (clock held /
2.4 GHz) is the rate delivered.  vector issue share = 4 * SQ_ACTIVE_INST_VALU / (1024 * GRBM_GUI_ACTIVE / 8) (MFMA issues
count as vector instructions too)."""
import collections, csv, glob, sys

d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[r["Kernel_Name"]][r["Counter_Name"]] += 1
print("| kernel | launches | cycles / launch (GRBM_GUI_ACTIVE / 8) | SQ_VALU_MFMA_BUSY_CYCLES / launch | mfma pipe busy | vector issue share | wave occupancy (SQ_WAVE_CYCLES x4 / SIMD-cycles) |")
print("|---|---|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    launches = n[k]["GRBM_GUI_ACTIVE"]
    simd_cycles = 1024.0 * gui / 8.0
    rows.append((gui, k, launches, gui / 8.0 / launches, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / launches,
                 c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd_cycles, 4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / simd_cycles,
                 4.0 * c.get("SQ_WAVE_CYCLES", 0.0) / simd_cycles))
for gui, k, launches, cyc, mf, busy, valu, occ in sorted(rows, reverse=True)[:28]:
    print("| `%s` | %d | %.0f | %.0f | %.3f | %.3f | %.2f |" % (k.replace("void ", "").split("(")[0][:70], launches, cyc, mf, busy, valu, occ))
