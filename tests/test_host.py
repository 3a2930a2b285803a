"""CPU suite: the C ABI library loads and exports every declared symbol, structs match the header, error
behaviour without a GPU, host-side sharding logic incl. a world_size-2 gloo run.  No GPU compute."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fftup.h")


def _declared_symbols():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"FFTUP_API\s+[\w\s\*]+?\b(fftup_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vkresample_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names          # the Python binding covers the whole header


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors of fftup_config / fftup_info have the compiler's sizes and offsets."""
    from vkresample_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(textwrap.dedent("""
        #include <stdio.h>
        #include <stddef.h>
        #include "fftup.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu\\n", sizeof(fftup_config), offsetof(fftup_config, upscale),
                   offsetof(fftup_config, device), offsetof(fftup_config, ring), sizeof(fftup_info));
            printf("%zu %zu %zu\\n", offsetof(fftup_info, alg_bytes_per_frame), offsetof(fftup_info, device_name),
                   offsetof(fftup_info, kernel_names));
            return 0;
        }"""))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    a, b = subprocess.check_output([str(exe)]).decode().split("\n")[:2]
    a = [int(x) for x in a.split()]
    b = [int(x) for x in b.split()]
    assert a == [C.sizeof(_lib.Config), _lib.Config.upscale.offset, _lib.Config.device.offset, _lib.Config.ring.offset,
                 C.sizeof(_lib.Info)]
    assert b == [_lib.Info.alg_bytes_per_frame.offset, _lib.Info.device_name.offset, _lib.Info.kernel_names.offset]


def test_header_compiles_as_c_and_cxx(tmp_path):
    for comp, std, ext in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "cpp")):
        f = tmp_path / ("t." + ext)
        f.write_text('#include "fftup.h"\nint main(void){ return FFTUP_OK; }\n')
        subprocess.check_call([comp, std, "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(f),
                               "-o", str(tmp_path / "t.o")])


def test_error_paths_without_device():
    """argument validation happens before any device access; without a GPU the product refuses to run."""
    import vkresample_amd as v
    from vkresample_amd import _lib
    lib = _lib.load()
    assert lib.fftup_strerror(0) == b"success"
    assert b"unsupported size" in lib.fftup_strerror(2)
    assert lib.fftup_version().startswith(b"fftup")
    for kwargs, code in ((dict(width=2 * 11 * 64, height=64), 2), (dict(width=63, height=64), 1),
                         (dict(width=64, height=64, precision=3), 3), (dict(width=64, height=64, upscale=0.5), 1),
                         # rows too long for the LDS are no errors any more (round 4: four steps through HBM) -- without a device
                         # such plans get as far as every other valid one: FFTUP_E_NO_DEVICE
                         (dict(width=4096, height=64, precision=1), 4), (dict(width=9216, height=64), 4), (dict(width=8064, height=64), 4)):
        with pytest.raises(v.FftupError) as e:
            v.Upscaler(**kwargs)
        assert e.value.code == code, kwargs
    h = C.c_void_p()
    cfg = _lib.Config(64, 32, 4, 2.0, 0, 0.2, 0, 0, 1)          # 4 channels
    assert lib.fftup_plan_create(C.byref(h), C.byref(cfg)) == 1
    assert lib.fftup_plan_create(None, C.byref(cfg)) == 1
    if v.device_count() == 0:
        with pytest.raises(v.FftupError) as e:
            v.Upscaler(64, 32)
        assert e.value.code == 4                                   # FFTUP_E_NO_DEVICE: no CPU fallback
    assert lib.fftup_execute(None, 1, None) == 1
    assert lib.fftup_submit_rgb8(None, None, 0, None, 0, None) == 1
    assert lib.fftup_wait(None, 0) == 1 and lib.fftup_drain(None) == 1
    assert lib.fftup_submit_png(None, None, 0, None, 0, None) == 1 and lib.fftup_wait_png(None, 0, None, 0, None) == 1 and lib.fftup_png_bound(None) == 0
    lib.fftup_host_free(None)                                      # no-op
    lib.fftup_plan_destroy(None)                                   # no-op


def test_product_does_not_touch_the_oracle():
    """the shipped path (package + csrc + CLI) never references oracle/ (tests, smoke and bench's cpu_baseline may)."""
    pkg = os.path.join(ROOT, "vkresample_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(base, f), errors="replace").read()
                assert "oraclelib" not in txt and "libfftup_oracle" not in txt and "orc_" not in txt, f
    ldd = subprocess.check_output(["ldd", os.path.join(pkg, "libfftup.so")]).decode()
    assert "oracle" not in ldd


def test_reference_stripe_semantics():
    from vkresample_amd.shard import frames_for_rank, local_frame_count
    for n, t in ((512, 8), (10, 3), (7, 8), (1, 1), (9, 4)):
        seen = []
        for r in range(t):
            fr = frames_for_rank(n, t, r)
            assert len(fr) == local_frame_count(n, t, r)
            assert all(f % t == r for f in fr)
            seen += fr
        assert sorted(seen) == list(range(n))                      # every frame exactly once
    assert frames_for_rank(10, 3, 0) == [0, 3, 6, 9] and frames_for_rank(10, 3, 2) == [2, 5, 8]


WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from vkresample_amd import synth
from vkresample_amd.shard import frames_for_rank, reduce_summary
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
frames = frames_for_rank(11, world, rank)
chk = sum(int(synth.frame(k, 32, 16).astype(np.int64).sum()) for k in frames)
n, total, tmax = reduce_summary(dist, len(frames), chk, 0.5 + rank)
if rank == 0:
    print("RESULT", n, total, tmax)
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_gloo_sharding(tmp_path):
    """world_size 2 on CPU (gloo): the shards cover the job exactly once and the end-of-run reduction is right."""
    import numpy as np
    from vkresample_amd import synth
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29741", str(script)],
                                  env=env, stderr=subprocess.STDOUT, timeout=240).decode()
    line = [l for l in out.splitlines() if l.startswith("RESULT")][0].split()
    expect = sum(int(synth.frame(k, 32, 16).astype(np.int64).sum()) for k in range(11))
    assert int(line[1]) == 11 and int(line[2]) == expect and float(line[3]) == 1.5


QUEUE_WORKER = """
import os, sys, time
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from vkresample_amd import synth
from vkresample_amd.shard import FrameQueue, reduce_summary
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
store = FrameQueue.default_store(dist)
got = []
for step in range(2):                                   # two steps = two counters
    mine = []
    for (a, b) in FrameQueue(store, 37, chunk=5, key="step%d" % step):
        mine += list(range(a, b))
        time.sleep(0.002 * (1 + 3 * rank))              # rank 1 is four times slower: rank 0 must end up with more frames
    got.append(mine)
chk = sum(int(synth.frame(k, 32, 16).astype(np.int64).sum()) for k in got[1])
n, total, tmax = reduce_summary(dist, len(got[1]), chk, 0.5 + rank)
all_frames = [None] * world
dist.all_gather_object(all_frames, got)
if rank == 0:
    print("RESULT", n, total, [len(g[1]) for g in all_frames], sorted(sum((g[0] for g in all_frames), [])) == list(range(37)),
          sorted(sum((g[1] for g in all_frames), [])) == list(range(37)))
dist.barrier(); dist.destroy_process_group()
"""


def test_two_rank_gloo_frame_queue(tmp_path):
    """world_size 2 on CPU (gloo): the shared counter (shard.FrameQueue, chunks of 5 of 37 frames) hands every frame out exactly
    once per step, the faster rank gets more of them, and the end-of-run reduction gives the stripe's totals."""
    import numpy as np
    from vkresample_amd import synth
    script = tmp_path / "qworker.py"
    script.write_text(QUEUE_WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29743")
    out = subprocess.check_output([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                   "--master-addr", "127.0.0.1", "--master-port", "29743", str(script)],
                                  env=env, stderr=subprocess.STDOUT, timeout=240).decode()
    line = [l for l in out.splitlines() if l.startswith("RESULT")][0]
    f = line.split(None, 3)
    expect = sum(int(synth.frame(k, 32, 16).astype(np.int64).sum()) for k in range(37))
    assert int(f[1]) == 37 and int(f[2]) == expect
    counts = eval(f[3].split("]")[0] + "]")
    assert sum(counts) == 37 and counts[0] > counts[1], counts
    assert f[3].endswith("True True"), line


def test_bench_refuses_a_world_that_is_not_gpus():
    """`--gpus N` must be the number of ranks that really run: a torch.distributed.run environment of another size is an
    error (exit code 2), never a line with a different n_gpus"""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "--gpus 1 but WORLD_SIZE=2" in r.stderr


def test_bench_gpus_flag_starts_ranks_itself():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two ranks (here, without a HIP device, both end with the
    product's "no CPU path" message -- the point is that two processes with RANK 0 and 1 were started, not one)"""
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "3"], 29555)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--repeats", "1",
                        "--frames-per-step", "2"], env=env, capture_output=True, text=True, timeout=600)
    import vkresample_amd as v
    if v.device_count() == 0:
        assert r.returncode != 0 and r.stderr.count("needs a HIP device") >= 2, r.stderr[-2000:]
    else:
        assert r.returncode != 0 or json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 2


def test_static_counter_figures_belong_to_these_kernel_sources():
    """bench.py prints HBM bytes and vector-instruction counts from committed rocprofv3 --pmc runs (profiles/hbm_traffic.json) and
    a second roofline fraction from the committed --kernel-trace --stats summaries (profiles/kernel_stats_index.json): both carry
    the fingerprint of the kernel sources they were measured on; a kernel change without a refresh (tools/gpu_round_end.sh) fails
    here instead of leaving stale figures in the line"""
    import bench
    h = bench.kernel_sources_sha256()
    tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    keys = [k for k in tj if not k.startswith("_")]
    assert {"2048x1024_p0_planar", "2048x1024_p2_u8", "1920x1080_p0_planar", "2048x1024_p2_u8_u8out"} <= set(keys)
    idx = json.load(open(os.path.join(ROOT, "profiles", "kernel_stats_index.json")))
    for k, e in idx.items():
        assert os.path.exists(os.path.join(ROOT, e["file"])), k
    stale = [k for k in keys if tj[k]["_kernel_sources_sha256"] != h] + [k for k, e in idx.items() if e["kernel_sources_sha256"] != h]
    if stale:
        # reported, not fatal: under `pytest -x` a stale profile must not keep the rest of the suite from running;
        # bench.py marks such figures STALE in its own line
        pytest.xfail("committed counter / trace summaries were measured on other kernel sources (%s): re-run tools/gpu_round_end.sh" % ", ".join(sorted(set(stale))))
    for k in keys:
        assert bench.traffic_is_current(os.path.join(ROOT, "profiles", "hbm_traffic.json"), k) is True
    us, f, fresh = bench.rocprof_kernel_us("2048x1024_p0_planar", "row_c2r_sharpen")
    assert 30 < us < 80 and fresh is True and f.startswith("profiles/")


def test_jit_check_compiles_without_a_device(tmp_path, monkeypatch):
    """csrc/jit.hpp: factorizations are picked and the translation unit is compiled by hipRTC for gfx950 with no GPU
    present; the second request is served from the on-disk cache; a size without a specialised factorization says so."""
    import ctypes as C
    import time
    from vkresample_amd import _lib
    monkeypatch.setenv("FFTUP_CACHE_DIR", str(tmp_path / "cache"))
    lib = _lib.load()
    buf = C.create_string_buffer(256)
    cases = [(640, 480, 0, "fused 8*10*16"),          # (the built-in wisdom table's pick for rows of 1280; the chooser's is 16*16*5)
             (896, 504, 0, "fused 16*16*7"), (1000, 1000, 2, "fused 8*5*5*10"), (1024, 768, 0, "row pow2/8"),
             (1200, 512, 0, "col pow2/8 digit-swap x256")]     # (k_col_v<4, 512>: kernels_dswap.hpp, the fifth embedded header)
    for (W, H, p, expect) in cases:
        rc = lib.fftup_jit_check(W, H, 2, p, None, buf, 256)
        assert rc == 0, lib.fftup_last_error().decode()
        assert expect in buf.value.decode(), buf.value
    files = list((tmp_path / "cache").glob("*.fjit"))
    # two code objects per plan: row + column kernels, fused C2R+sharpen + stand-alone C2R
    assert len(files) == 2 * len(cases) and all(f.stat().st_size > 10000 for f in files)
    # a plan of another height shares the second one (it depends on the output row length only)
    assert lib.fftup_jit_check(896, 648, 2, 0, None, buf, 256) == 0
    assert len(list((tmp_path / "cache").glob("*.fjit"))) == 2 * len(cases) + 1
    # ... but not with a plan of the same row length and another factor (960 x 2 = 640 x 3 = 1920: the instantiations differ)
    n0 = len(list((tmp_path / "cache").glob("*.fjit")))
    assert lib.fftup_jit_check(960, 540, 2, 0, None, buf, 256) == 0 and lib.fftup_jit_check(640, 360, 3, 0, None, buf, 256) == 0
    assert len(list((tmp_path / "cache").glob("*.fjit"))) == n0 + 4
    assert lib.fftup_jit_check(2000, 1250, 2, 0, None, buf, 256) == 0 and "row 8*5*5*10" in buf.value.decode()       # N-stage kernels
    assert lib.fftup_jit_check(4000, 3000, 2, 0, None, buf, 256) == 0 and "(2 columns)" in buf.value.decode()      # long columns
    assert lib.fftup_jit_check(2450, 1080, 2, 0, None, buf, 256) == 2          # FFTUP_E_UNSUPPORTED_SIZE: the generic kernels run it
    assert lib.fftup_jit_check(640, 480, 2, 1, None, buf, 256) == 3            # FFTUP_E_UNSUPPORTED_PRECISION
    # a pinned factorization that does not multiply to the size is ignored; a valid one is used
    monkeypatch.setenv("FFTUP_LIBRARY", _lib.KNOBS_LIB_PATH)       # the FFTUP_EXPERIMENT parser exists in the test build only
    lib = _lib.load()
    monkeypatch.setenv("FFTUP_EXPERIMENT", "jit_row=5,8,16")
    assert lib.fftup_jit_check(640, 480, 2, 0, None, buf, 256) == 0 and "row 5*8*16" in buf.value.decode()
    monkeypatch.setenv("FFTUP_EXPERIMENT", "jit_row=5,8,8")
    assert lib.fftup_jit_check(640, 480, 2, 0, None, buf, 256) == 0 and "row 10*8*8" in buf.value.decode()


def test_shipping_library_has_no_experiment_parser(monkeypatch):
    """FFTUP_EXPERIMENT is a test knob.  The shipping libfftup.so does not parse it (the same pin that changes
    libfftup_knobs.so's choice leaves the product's untouched); nothing else in csrc/ reads the variable."""
    from vkresample_amd import _lib
    buf = C.create_string_buffer(256)
    monkeypatch.setenv("FFTUP_EXPERIMENT", "jit_row=5,8,16")
    monkeypatch.delenv("FFTUP_LIBRARY", raising=False)
    lib = _lib.load()
    assert lib.fftup_jit_check(640, 480, 2, 0, b"", buf, 256) == 0 and "row 5*8*16" not in buf.value.decode()
    monkeypatch.setenv("FFTUP_LIBRARY", _lib.KNOBS_LIB_PATH)
    assert _lib.load().fftup_jit_check(640, 480, 2, 0, b"", buf, 256) == 0 and "row 5*8*16" in buf.value.decode()
    csrc = os.path.join(ROOT, "vkresample_amd", "csrc")
    readers = [f for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".hpp")) and 'getenv("FFTUP_EXPERIMENT")' in open(os.path.join(csrc, f)).read()]
    assert readers == ["jit.cpp"]
    src = open(os.path.join(csrc, "jit.cpp")).read()
    k = src.index("#ifdef FFTUP_TEST_KNOBS")
    assert k < src.index('getenv("FFTUP_EXPERIMENT")') < src.index("#else", k)


def test_jit_cache_rejects_torn_files_and_coalesces_concurrent_compiles(tmp_path):
    """(a) a cache file with damaged bytes (a torn write, another process still writing) must not load -- the
    header carries a checksum; the plan is compiled again and the file replaced; (b) threads that ask for the same
    translation unit at the same moment (the CLI's -numthreads mode) get ONE compilation, not one each, and no two writers
    share a temporary file.  Each part in a process of its own (the in-memory table must not answer)."""
    import subprocess
    import sys
    cache = tmp_path / "cache"
    # (AMD_COMGR_CACHE=0: hipRTC's own code cache would make a compilation as quick as a load, and the timings below blind)
    env = dict(os.environ, FFTUP_CACHE_DIR=str(cache), AMD_COMGR_CACHE="0")
    check = ("import ctypes as C, sys, time; sys.path.insert(0, %r); from vkresample_amd import _lib; lib = _lib.load(); "
             "buf = C.create_string_buffer(256); t = time.time(); rc = lib.fftup_jit_check(720, 576, 2, 0, None, buf, 256); "
             "print(rc, '%%.2f' %% (time.time() - t))" % ROOT)
    r = subprocess.run([sys.executable, "-c", check], env=env, capture_output=True, text=True, timeout=300)
    assert r.stdout.split()[0] == "0", r.stdout + r.stderr
    files = sorted(cache.glob("*.fjit"))
    assert len(files) == 2 and not list(cache.glob("*.fjit.*"))                   # no temporaries left behind
    good = [f.read_bytes() for f in files]
    assert all(g[:6] == b"FJIT2\n" for g in good)
    # (a) flip one byte in the middle of each code object
    for f, g in zip(files, good):
        f.write_bytes(g[:len(g) // 2] + bytes([g[len(g) // 2] ^ 0x55]) + g[len(g) // 2 + 1:])
    r = subprocess.run([sys.executable, "-c", check], env=env, capture_output=True, text=True, timeout=300)
    rc, secs = r.stdout.split()
    assert rc == "0" and float(secs) > 0.3, r.stdout + r.stderr                   # compiled again, not loaded
    def whole(raw):                                                                # FJIT2: magic, fnv1a-64 of the rest, payload
        h = 1469598103934665603
        for ch in raw[14:]:
            h = ((h ^ ch) * 1099511628211) & (2 ** 64 - 1)
        return raw[:6] == b"FJIT2\n" and int.from_bytes(raw[6:14], "little") == h
    assert all(whole(g) for g in good) and all(whole(f.read_bytes()) for f in files)       # and the files are whole again
    r = subprocess.run([sys.executable, "-c", check], env=env, capture_output=True, text=True, timeout=300)
    assert r.stdout.split()[0] == "0" and float(r.stdout.split()[1]) < 0.3, r.stdout         # now served from the cache
    # (b) four threads, same plan, cold cache: about the time of one compilation
    for f in files:
        f.unlink()
    conc = ("import ctypes as C, sys, time, threading; sys.path.insert(0, %r); from vkresample_amd import _lib; lib = _lib.load();\n"
            "def one():\n    buf = C.create_string_buffer(256); assert lib.fftup_jit_check(720, 576, 2, 0, None, buf, 256) == 0\n"
            "t = time.time(); ths = [threading.Thread(target=one) for _ in range(4)]; [x.start() for x in ths]; [x.join() for x in ths]; "
            "print('%%.2f' %% (time.time() - t))" % ROOT)
    r1 = subprocess.run([sys.executable, "-c", check], env=dict(env, FFTUP_CACHE_DIR=str(tmp_path / "c2")), capture_output=True, text=True, timeout=300)
    single = float(r1.stdout.split()[1])
    r = subprocess.run([sys.executable, "-c", conc], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert float(r.stdout.split()[0]) < 2.2 * single + 0.5, (r.stdout, single)     # (four compilations in a row would be 4 x)
    assert len(list(cache.glob("*.fjit"))) == 2 and not list(cache.glob("*.fjit.*"))


def test_jit_chooser_invariants_over_all_sizes():
    """Every factorization the plan-time chooser (csrc/jit.hpp) can hand out, for all even 2,3,5,7-smooth widths and a
    spread of heights up to 4096 and the factors 1.5 ... 8: radices multiply to the length, the workgroup holds the first
    and the last stage, at most 16 points per thread, the fused kernel's first radix is a multiple of 2u, LDS within
    160 KB, at most 1024 threads.  (No compilation: arch = "".)"""
    import ctypes as C
    import re
    import numpy as np
    from vkresample_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(512)
    smooth = sorted({2 ** a * 3 ** b * 5 ** c * 7 ** d for a in range(1, 13) for b in range(6) for c in range(5) for d in range(4)
                     if 64 <= 2 ** a * 3 ** b * 5 ** c * 7 ** d <= 4096})
    radices = {2, 3, 4, 5, 7, 8, 9, 10, 12, 14, 15, 16}

    def prod(xs):
        p = 1
        for x in xs:
            p *= x
        return p

    def parse(part):       # "8*5*5*10 x256" -> ([8,5,5,10], 256)
        m = re.match(r"([\d*]+) x(\d+)", part)
        return [int(x) for x in m.group(1).split("*")], int(m.group(2))
    seen = 0
    for i, W in enumerate(smooth):
        H = smooth[(i * 7 + 3) % len(smooth)]
        for u in (1.125, 1.25, 1.5, 1.75, 1.875, 2.0, 2.5, 3.0, 4.0, 5.0, 7.0, 8.0, float(np.float32(4 / 3)), 1.6, 1.4, float(np.float32(5 / 3))):
            rc = lib.fftup_jit_check(W, H, u, 0, b"", buf, 512)
            assert rc in (0, 2, 1), (W, H, u, rc)
            if rc != 0:
                continue
            seen += 1
            d = buf.value.decode()
            UW, UH = int(np.float32(u) * np.float32(W)), int(np.float32(u) * np.float32(H))      # (float arithmetic, as VR:1417)
            DD = next(dd for dd in (1, 2, 4, 3, 5, 7) if abs(2 * dd * u - round(2 * dd * u)) < 1e-5 * 2 * dd * u)       # the factor as D / (2 DD)
            D = round(2 * DD * u)
            assert UW % 4 == 0 and UW <= 8192
            m = re.search(r"row (.*?), col (.*?), fused (.*?) \((\d+) B LDS", d)
            assert m, d
            row, col, fused, lds = m.group(1), m.group(2), m.group(3), int(m.group(4))
            assert lds <= 160 * 1024
            if not row.startswith(("pow2", "generic")):
                r, t = parse(row)
                assert prod(r) == W and set(r) <= radices and t <= 1024 and t >= W // r[0] and t >= W // r[-1], d
            cols = 2 if "(2 columns)" in col else 4
            col = col.replace(" (2 columns)", "")
            if not col.startswith("pow2"):
                if "->" in col:                                     # half-integer factor: forward H, inverse uH
                    f, rest = col.split(" -> ")
                    fr = [int(x) for x in f.split("*")]
                    ir, t = parse(rest)
                    assert prod(fr) == H and prod(ir) == UH and set(fr) | set(ir) <= radices, d
                    assert t <= 1024 and t >= cols * max(H // fr[0], H // fr[-1], UH // ir[0], UH // ir[-1]), d
                else:
                    r, t = parse(col)
                    assert prod(r) == H and set(r) <= radices and t <= 1024 and t >= cols * max(H // r[0], H // r[-1]), d
            if not fused.startswith("pow2"):
                r, t = parse(fused)
                assert prod(r) == UW and set(r) <= radices and (r[0] * DD) % D == 0 and t <= 1024 and t % 64 == 0, d
                assert t >= UW // r[0] and t >= UW // r[-1], d
                assert max(-(-(UW // q) // t) * q for q in r) <= 16, d          # points per thread
    assert seen > 400


def test_without_the_runtime_compiler_the_library_says_so(tmp_path):
    """libhiprtc.so is dlopen'ed: a machine without it keeps working on the size-generic kernels; fftup_jit_check reports
    it (FFTUP_E_HIP, 'hipRTC ... not available') instead of failing to load the library."""
    import subprocess
    import sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r); from vkresample_amd import _lib; lib = _lib.load(); "
            "buf = C.create_string_buffer(256); rc = lib.fftup_jit_check(640, 480, 2.0, 0, None, buf, 256); "
            "print(rc, lib.fftup_last_error().decode())" % ROOT)
    env = dict(os.environ, FFTUP_HIPRTC_LIB=str(tmp_path / "no_such_libhiprtc.so"), FFTUP_CACHE_DIR=str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    assert r.stdout.startswith("5 ") and "not available" in r.stdout, r.stdout


def test_jit_check_takes_any_float_factor():
    """fftup_jit_check (no device, no compilation) over factors that are whatever float a caller may pass -- just off an integer, off a
    ratio, irrational, huge: it answers 0 (a specialised plan exists: then the output sizes are exactly D / (2 DD) times the input
    sizes for the D, DD its description names), 1 (not a configuration at all) or 2 (the size-generic kernels run it); it never faults."""
    import ctypes as C
    import re
    import numpy as np
    from fractions import Fraction
    from vkresample_amd import _lib
    lib = _lib.load()
    buf = C.create_string_buffer(512)
    rng = np.random.default_rng(7)
    sizes = [64, 96, 100, 120, 128, 250, 256, 270, 480, 600, 720, 900, 1000, 1080, 1280, 1600, 1920, 2048]
    base = [1.0, 1.125, 1.2, 1.25, 4 / 3, 1.4, 1.5, 1.6, 5 / 3, 1.75, 1.8, 1.875, 2.0, 2.25, 2.4, 2.5, 8 / 3, 3.0, 3.5, 4.0, 5.0, 7.0, 8.0, 9.0, 16.0, np.pi, np.e]
    seen = 0
    for k in range(3000):
        W, H = int(rng.choice(sizes)), int(rng.choice(sizes))
        u = np.float32(rng.choice(base))
        if k % 3 == 1:
            u = np.nextafter(u, np.float32(0 if k % 2 else 100), dtype=np.float32)       # one float off
        elif k % 3 == 2:
            u = np.float32(u * (1 + rng.normal() * 1e-6))
        rc = lib.fftup_jit_check(W, H, C.c_float(float(u)), int(rng.choice([0, 2])), b"", buf, 512)
        assert rc in (0, 1, 2), (W, H, float(u), rc)
        if rc == 0:
            seen += 1
            uW, uH = int(np.float32(u) * np.float32(W)), int(np.float32(u) * np.float32(H))
            fr = Fraction(uW, W)
            assert Fraction(uH, H) == fr and fr.limit_denominator(16) == fr, (W, H, float(u), uW, uH)
            m = re.search(r"fused ([\d*]+|pow2/8)", buf.value.decode())
            assert m, buf.value
    assert seen > 300
