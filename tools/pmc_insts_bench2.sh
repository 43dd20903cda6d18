# Vector-instruction mix by type per train-step kernel (eager c2 bench under rocprofv3 --pmc, two passes), per wave.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_insts_b2; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT SQ_WAVES --output-format csv -d $O/a -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES --output-format csv -d $O/b -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --no_mode_sweep --graph 0 > /dev/null 2> $O/b.err
python - <<'PY'
import csv, glob, collections
for d in "ab":
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for f in glob.glob(f"gpurun_out/pmc_insts_b2/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=(r["Kernel_Name"].split("(")[0][:44], r["Grid_Size"]); acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("SQ_WAVES",0)):
        if ("mlp3" not in k[0] and "wgrad" not in k[0]) or len(n[k])<20: continue
        w=c["SQ_WAVES"]
        print(d, f"{k[0]:44s} {k[1]:>7s}", "  ".join(f"{x[9:]}={v/w:.0f}" for x,v in sorted(c.items()) if x!="SQ_WAVES"))
PY
