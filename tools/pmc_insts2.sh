# Co-execution, LDS-conflict, instruction-fetch and level counters of the bare twin-Q forward (three rocprofv3 --pmc passes).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_insts2; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/a -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/b -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/b.err
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_CYCLES SQ_LEVEL_WAVES --output-format csv -d $O/c -- python3 tools/probe/micro_pipe.py 15360 > /dev/null 2> $O/c.err
python - <<'PY'
import csv, glob, collections
for d in "abc":
    acc=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob(f"gpurun_out/pmc_insts2/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mlp3_fwd_bf" not in r["Kernel_Name"]: continue
            k=r["Counter_Name"]; acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,(v,n) in sorted(acc.items()): print(d, k, "per launch", v/max(n,1), "launches", n)
PY
