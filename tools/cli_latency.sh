#!/bin/bash
# wall time of ONE invocation of the drop-in CLI (process start, HIP bring-up, plan, PNG in, frame, PNG out), what a user of the
# reference's `VkResample -i in.png -o out.png -u 2` waits for:   gpurun -- tools/cli_latency.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
T=$(mktemp -d); cd $T
python - <<EOF
import sys; sys.path.insert(0, "$R")
from PIL import Image
from vkresample_amd import synth
for (W, H) in ((1920, 1080), (2048, 1024), (1280, 720)):
    Image.fromarray(synth.frame(3, W, H, "N")).save("in_%dx%d.png" % (W, H))
EOF
export FFTUP_CACHE_DIR=$T/cache
for s in 2048x1024:2 1920x1080:2 1280x720:1.5 1920x1080:1.3333334; do
  sz=${s%%:*}; u=${s##*:}
  for rep in 1 2 3; do
    t0=$(date +%s.%N)
    $R/vkresample_amd/vkresample -i in_$sz.png -o out.png -u $u -n 1 > log.txt 2>&1
    t1=$(date +%s.%N)
    echo "$sz -u $u run $rep: $(python -c "print('%.3f' % ($t1 - $t0))") s wall | $(grep -o 'Time: [0-9.]* ms' log.txt) | $(grep -io 'total time[^,]*' log.txt | head -1)"
  done
done
rm -rf $T
