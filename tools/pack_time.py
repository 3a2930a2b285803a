"""k_pack_u8 / k_unpack_u8 under rocprofv3 --kernel-trace --stats: python tools/pack_time.py  (20 blocking downloads of a 4096x2048 image, fp32 and -p 2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vkresample_amd as v
from vkresample_amd import synth
for p in (0, 2):
    with v.Upscaler(2048, 1024, 2.0, p) as up:
        up.upload_rgb8(synth.frame(1, 2048, 1024))
        up.execute(1)
        for _ in range(20):
            up.download_rgb8()
