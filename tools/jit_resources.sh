#!/bin/bash
# registers / scratch / LDS of the kernels of a run-time specialised plan, compiled offline:
#   tools/jit_resources.sh <W> <H> [precision] [upscale]      (FFTUP_EXPERIMENT jit_row / jit_col / jit_fused pins are honoured)
W=$1; H=$2; P=${3:-0}; U=${4:-2}
cd "$(dirname "$0")/.."
FFTUP_CACHE_DIR=/tmp/jit_res_cache FFTUP_LIBRARY=$(dirname $0)/../vkresample_amd/libfftup_knobs.so FFTUP_EXPERIMENT="${FFTUP_EXPERIMENT:+$FFTUP_EXPERIMENT;}jit_dump=/tmp/jit_res_$$.hip" python - <<PY
import ctypes as C, sys
sys.path.insert(0, ".")
from vkresample_amd import _lib
lib = _lib.load()
buf = C.create_string_buffer(512)
import os, shutil
shutil.rmtree("/tmp/jit_res_cache", ignore_errors=True)
rc = lib.fftup_jit_check($W, $H, float($U), $P, None, buf, 512)
print(rc, buf.value.decode(), lib.fftup_last_error().decode()[:500] if rc else "")
PY
for part in rowcol fused; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on --cuda-device-only -c -Ivkresample_amd/csrc -o /tmp/jit_res_$$.o /tmp/jit_res_$$.hip.$part.hip -Rpass-analysis=kernel-resource-usage 2>&1; done | python3 -c '
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m:
        cur={"name":m.group(1)}; rows.append(cur); continue
    for key,pat in (("vgpr",r" VGPRs: (\d+)"),("scratch",r"ScratchSize \[bytes/lane\]: (\d+)"),("occ",r"Occupancy \[waves/SIMD\]: (\d+)"),("lds",r"LDS Size \[bytes/block\]: (\d+)")):
        m=re.search(pat,l)
        if m and cur is not None: cur[key]=m.group(1)
    if "error" in l: print(l.rstrip())
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip().split("(")[0].replace("void fftup::","")
    print("%-70s vgpr %4s scratch %3s occ %2s lds %6s"%(name[:70],r.get("vgpr"),r.get("scratch"),r.get("occ"),r.get("lds")))
'
rm -f /tmp/jit_res_$$.*
