#!/bin/bash
# prints per-kernel VGPR/SGPR/LDS/occupancy of the HIP library (compile only)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -fPIC -shared -fvisibility=hidden -o /tmp/_ru.so vkresample_amd/csrc/fftup.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    for key,pat in (("sgpr",r" SGPRs: (\d+)"),("vgpr",r" VGPRs: (\d+)"),("agpr",r"AGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)")):
        m=re.search(pat,l)
        if m and cur is not None: cur[key]=m.group(1)
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip().split("(")[0].replace("void fftup::","")
    print("%-52s vgpr %4s sgpr %4s scratch %3s occ %2s lds %6s"%(name[:52],r.get("vgpr"),r.get("sgpr"),r.get("scratch"),r.get("occ"),r.get("lds")))
'
