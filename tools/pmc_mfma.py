"""MFMA-pipe utilisation per kernel from one rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_m -- python3 bench.py ...
    python tools/pmc_mfma.py gpurun_out/pmc_m profiles/<tag>_pmc_mfma.json
SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe is busy (32 per
v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32: MI355X guide, cycle constants); GRBM_GUI_ACTIVE is the dispatch's
cycles summed over the 8 XCDs.  utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024) = the fraction of the launch's SIMD
cycles, at the clock the chip actually held, in which the matrix pipe was busy."""
import collections, csv, glob, json, os, sys

d, out = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0], r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        did = (key, r["Dispatch_Id"])
        if did not in seen:
            seen.add(did); n[key] += 1
rows = []
for key, c in acc.items():
    if "mobody" not in key[0]:
        continue
    gui, busy, sq = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CYCLES", 0.0)
    rows.append(dict(kernel=key[0], grid=key[1], launches=n[key], mfma_busy_cycles_per_launch=busy / n[key],
                     gui_active_per_launch=gui / n[key], sq_busy_cycles_per_launch=sq / n[key],
                     mfma_util=busy / (gui / 8 * 1024) if gui else None))
rows.sort(key=lambda e: -e["mfma_busy_cycles_per_launch"] * e["launches"])
json.dump(rows, open(out, "w"), indent=1)
for e in rows[:16]:
    print(f"{e['kernel'][:60]:60s} grid {e['grid']:>8s} x{e['launches']:<4d} MFMA busy {e['mfma_busy_cycles_per_launch']:12.0f} cyc  "
          f"util {100 * (e['mfma_util'] or 0):5.1f} %")
