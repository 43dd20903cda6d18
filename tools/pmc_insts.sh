# Instruction / wait counters of the bare twin-Q forward (tools/probe/micro_pipe.py 15360) in three rocprofv3 --pmc passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_insts; mkdir -p $O
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $O/a -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_MFMA --output-format csv -d $O/b -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/b.err
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $O/c -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/c.err
python - <<'PY'
import csv, glob, collections
for d in "abc":
    acc=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f"gpurun_out/pmc_insts/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp3_fwd_bf" not in r["Kernel_Name"]: continue
            k=r["Counter_Name"]; acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,(v,n) in sorted(acc.items()): print(d, k, "per launch", v/max(n,1), "launches", n)
PY
tail -2 $O/a.err $O/b.err $O/c.err | grep -i "error\|invalid" | head
