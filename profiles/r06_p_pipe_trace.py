#!/usr/bin/env python3
"""Timeline of the plane pipeline: run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/pipe_trace.py run`,
then `python tools/pipe_trace.py show <dir>` prints begin / end of every kernel of a few consecutive frames relative to the frame's first launch."""
import csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import vkresample_amd as v
    from vkresample_amd import synth
    W, H = 2048, 1024
    with v.Upscaler(W, H, 2.0, 0, 0.2, 0, 0, 1) as up:
        up.upload_rgb8(synth.frame(0, W, H, "U"))
        up.execute(50)
        print("us per iteration:", up.execute(200) * 1e3)
else:
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "fftup" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-60:]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        name = r["Kernel_Name"].replace("void fftup::", "").split("<")[0].split("(")[0]
        print("%-18s queue %-3s %8.1f .. %8.1f us  (%.1f)" % (name, r.get("Queue_Id", "?"), (int(r["Start_Timestamp"]) - t0) * 1e-3, (int(r["End_Timestamp"]) - t0) * 1e-3,
                                                        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
