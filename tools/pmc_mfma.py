"""MFMA-pipe utilisation per kernel from one rocprofv3 counter pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d gpurun_out/pmc_m -- python3 bench.py ...
    python tools/pmc_mfma.py gpurun_out/pmc_m profiles/<tag>_pmc_mfma.json
SQ_VALU_MFMA_BUSY_CYCLES counts, summed over the chip's 1024 SIMDs, the cycles a SIMD's matrix pipe is busy (32 per
v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32: MI355X guide, cycle constants); GRBM_GUI_ACTIVE is the dispatch's
cycles summed over the 8 XCDs.  Two denominators:
  mfma_util     = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024): against the dispatch window the profiler sees.  Under counter
                  collection that window includes the dispatch's start / stop overhead (~12 us around a 30 us kernel: the
                  forward's window is 88 900 cycles for a 30.4 us kernel), so short kernels read low;
  mfma_util_cu  = MFMA_BUSY / (SQ_BUSY_CU_CYCLES * 4): against the cycles the CUs actually held waves (4 SIMDs per CU), at the
                  clock the chip held -- the fraction of a busy SIMD's time its matrix pipe was busy.
valu_util_cu = 4 * SQ_ACTIVE_INST_VALU / (SQ_BUSY_CU_CYCLES * 4) (the counter ticks once per 4-cycle vector instruction slot);
coexec = SQ_VALU_MFMA_COEXEC_CYCLES / MFMA_BUSY: the share of matrix-busy cycles in which a vector instruction also ran."""
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
    gui, busy, cu = c.get("GRBM_GUI_ACTIVE", 0.0), c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("SQ_BUSY_CU_CYCLES", 0.0)
    valu, co = c.get("SQ_ACTIVE_INST_VALU", 0.0), c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0)
    rows.append(dict(kernel=key[0], grid=key[1], launches=n[key], mfma_busy_cycles_per_launch=busy / n[key],
                     gui_active_per_launch=gui / n[key], busy_cu_cycles_per_launch=cu / n[key],
                     mfma_util=busy / (gui / 8 * 1024) if gui else None, mfma_util_cu=busy / (cu * 4) if cu else None,
                     valu_util_cu=valu / cu if cu else None, coexec=co / busy if busy else None))
rows.sort(key=lambda e: -e["mfma_busy_cycles_per_launch"] * e["launches"])
json.dump(rows, open(out, "w"), indent=1)
for e in rows[:16]:
    print(f"{e['kernel'][:52]:52s} grid {e['grid']:>8s} x{e['launches']:<4d} MFMA busy {e['mfma_busy_cycles_per_launch']:11.0f} cyc  "
          f"of window {100 * (e['mfma_util'] or 0):5.1f} %  of busy-CU cycles {100 * (e['mfma_util_cu'] or 0):5.1f} %  "
          f"VALU {100 * (e['valu_util_cu'] or 0):5.1f} %  coexec {100 * (e['coexec'] or 0):4.1f} %")
