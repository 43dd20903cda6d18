# Vector-instruction mix by type of the bare twin-Q forward (two rocprofv3 --pmc passes), per wave.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_insts3; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT --output-format csv -d $O/a -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F16 --output-format csv -d $O/b -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/b.err
python - <<'PY'
import csv, glob, collections
for d in "ab":
    acc=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f"gpurun_out/pmc_insts3/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp3_fwd_bf" not in r["Kernel_Name"]: continue
            k=r["Counter_Name"]; acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,(v,n) in sorted(acc.items()): print(d, k, "per wave", v/max(n,1)/3840)
PY
