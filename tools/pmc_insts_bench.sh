# per-kernel instruction counts of the c2 train step (eager bench under rocprofv3 --pmc): bash tools/pmc_insts_bench.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_insts_bench; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/a -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/a.err
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
for f in glob.glob("gpurun_out/pmc_insts_bench/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=(r["Kernel_Name"].split("(")[0][:46], r["Grid_Size"]); acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_INSTS_VALU",0)):
    if "mobody" not in k[0] or len(n[k])<20: continue
    w=c["SQ_WAVES"]
    print(f"{k[0]:46s} grid {k[1]:>7s} x{len(n[k]):<4d} per wave: VALU {c['SQ_INSTS_VALU']/w:7.0f} (MFMA {c['SQ_INSTS_MFMA']/w:5.0f})  SALU {c['SQ_INSTS_SALU']/w:6.0f}  LDS {c['SQ_INSTS_LDS']/w:5.0f}  VMEM rd {c['SQ_INSTS_VMEM_RD']/w:5.0f} wr {c['SQ_INSTS_VMEM_WR']/w:5.0f}  waves {w/len(n[k]):6.0f}")
PY
