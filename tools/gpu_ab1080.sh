#!/bin/bash
cp vkresample_amd/libfftup.so /tmp/libfftup_base.so
for rep in 1 2; do for lib in /tmp/libfftup_base.so "$@"; do cp $lib vkresample_amd/libfftup.so; echo -n "$lib: "; python - <<PY
import vkresample_amd as v
from vkresample_amd import synth
up=v.Upscaler(1920,1080,2.0,0,0.2,0,0,4)
for s in range(4): up.upload_rgb8(synth.frame(s,1920,1080),slot=s)
up.execute_ring(64,0); ms=up.execute_ring(256,0)/256
print("ms/frame %.4f"%ms, dict(zip(up.kernel_names,[round(x*1e3,1) for x in up.profile_kernels(20)])))
PY
done; done
cp /tmp/libfftup_base.so vkresample_amd/libfftup.so
