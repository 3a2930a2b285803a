#!/bin/bash
# host-streamed queue: tests + bench lines (PCIe-inclusive; not the headline)
mkdir -p gpurun_out/stream
timeout 600 python -m pytest tests -m gpu -q -k "host_streamed" 2>&1 | tail -5
for ring in 2 4 8; do
  timeout 300 python bench.py --host-streamed --ring $ring --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/stream/bench_ring$ring.json 2> gpurun_out/stream/err_ring$ring.log
  python - <<PY
import json
l=json.load(open("gpurun_out/stream/bench_ring$ring.json"))
print("ring $ring: %.0f fps  %.3f ms/frame  PCIe %.1f GB/s" % (l["value"], l["ms_per_frame"], l["pcie_GBps"]))
PY
done
timeout 300 python bench.py --host-streamed --ring 4 --fuse-u8 --precision 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/stream/bench_p2.json 2>/dev/null
python -c "
import json
l=json.load(open('gpurun_out/stream/bench_p2.json'))
print('p2 fused-u8 ring 4: %.0f fps  %.3f ms/frame  PCIe %.1f GB/s' % (l['value'], l['ms_per_frame'], l['pcie_GBps']))"
