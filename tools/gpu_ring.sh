#!/bin/bash
for r in 4 8 16; do
  python bench.py --no-cpu-baseline --ring $r > /tmp/ring_$r.json 2>/tmp/ring_$r.err
  python - <<PY
import json
d=json.load(open("/tmp/ring_$r.json")); print("ring", $r, round(d["value"]), "fps", round(d["ms_per_frame"]*1e3,2), "us")
PY
done
