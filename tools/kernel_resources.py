"""VGPR / scratch / occupancy of every kernel of a translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py csrc/file.hip [...]   (cross-compiles; no GPU needed)"""
import re
import subprocess
import sys

for src in sys.argv[1:]:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", src, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    for b in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = b.split()[0]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(mobody::.*", "", dem).replace("void mobody::", "")
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "-"])[1]
        vg, ag, sc = g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]")
        occ, lds = g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{dem[:64]:64s} VGPR {vg:>4s} AGPR {ag:>4s} scratch {sc:>5s} occ {occ} lds {lds}")
