"""HBM-side traffic per kernel launch from two rocprofv3 counter passes (FETCH_SIZE and WRITE_SIZE cannot share a pass:
MI355X_MICROARCH.md "rocprofv3 PMC slots").  Collect with

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --graph 0
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -- python3 bench.py --steps 40 --warmup 3 --no_cpu_baseline --graph 0

then  python tools/pmc_traffic.py gpurun_out/pmc_f gpurun_out/pmc_w profiles/<tag>_pmc_traffic.json [config] [mfma]
The summary records the source fingerprint of the build (bench.src_fingerprint): bench.py only quotes it for the same build.
Values are KiB as rocprofv3 reports them (fetch_kb_raw is NOT yet doubled; gfx950 tallies 128-B read requests at 64 B)."""
import collections, csv, glob, json, os, sys


def collect(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            key = (name, r["Grid_Size"])
            acc[key][0] += float(r["Counter_Value"]); acc[key][1] += 1
    return acc


fd, wd, out = sys.argv[1:4]
config = sys.argv[4] if len(sys.argv) > 4 else "c2"
mfma = sys.argv[5] if len(sys.argv) > 5 else "f16x2"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
F, W = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
rows = []
for key in sorted(set(F) | set(W)):
    f, w = F.get(key), W.get(key)
    rows.append(dict(kernel=key[0], grid=key[1], launches=(f or w)[1],
                     fetch_kb_raw=f[0] / f[1] if f else None, write_kb=w[0] / w[1] if w else None))
rows.sort(key=lambda e: -((e["fetch_kb_raw"] or 0) * 2 + (e["write_kb"] or 0)) * e["launches"])
json.dump(dict(src_fingerprint=bench.src_fingerprint(), config=config, mfma=mfma, kernels=rows), open(out, "w"), indent=1)
for e in rows[:24]:
    if "mobody" in e["kernel"]:
        print(f"{e['kernel'][:56]:56s} grid {e['grid']:>8s} x{e['launches']:<4d} fetch(2x) {2*(e['fetch_kb_raw'] or 0)/1024:8.1f} MB  write {(e['write_kb'] or 0)/1024:8.1f} MB")
