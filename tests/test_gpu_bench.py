"""GPU: bench.py's contract -- one JSON line with the fields the driver and the judge read, single rank and two ranks
(two processes sharing the one GPU of the box over the gloo backend; on a multi-GPU node the same launch line runs one
rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


@pytest.fixture(scope="module")
def bench_line():
    """ONE run of `python bench.py` as the driver runs it (no cache dir pinned), shared by the tests below: the line's contract is
    asserted hard, what depends on this box's speed, on rocprofv3 / RCCL coming up, or on the committed profiles being current is
    reported as an expected failure with its reason (pytest -x does not stop there: the parity record stands either way)."""
    env = {k: v for k, v in os.environ.items() if k != "FFTUP_CACHE_DIR"}
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--repeats", "3", "--frames-per-step", "64",
                        "--cpu-frames", "1", "--profile-iters", "5"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r.stdout)


def _soft(cond, reason):
    if not cond:
        pytest.xfail(reason)


def test_bench_single_rank_line(bench_line):
    """the contract of the one JSON line: metric, value, roofline, cpu_baseline, config"""
    d = bench_line
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "frames/s" and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["value"] > 1000 and abs(d["value"] - 2 * 64 / d["timed_region_s_median"]) < 1e-6 * d["value"]
    assert len(d["region_s"]) == 3 and "workload" in d["config"] and "model" not in d["config"]
    ro = d["roofline"]
    assert ro["bound"] == "hbm" and ro["peak"] == 8000.0 and ro["unit"] == "GB/s" and abs(ro["frac"] - ro["achieved"] / 8000.0) < 1e-9
    assert ro["kernel"] == "row_c2r_sharpen" and 0.2 < ro["frac"] < 1.0
    assert ro["traffic"] is None or 5e7 < ro["traffic"] < 4e8
    assert d["B_min"] == 3.0 * (2048 * 1024 * 4 + 4096 * 2048 * 4) and d["frame_alg_bytes"] > d["B_min"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert "built-in" in d["config"]["wisdom"]               # no tuner findings of earlier runs on this machine steer the run
    # socket power and shader clock during the timed regions (best effort: None where the driver's hwmon files are missing)
    pw = d["power"]
    assert set(("socket_power_w_median", "power_cap_w", "sclk_mhz_median")) <= set(pw)
    assert pw["socket_power_w_median"] is None or 100 < pw["socket_power_w_median"] < 2500
    assert pw["sclk_mhz_median"] is None or 100 < pw["sclk_mhz_median"] < 3500


def test_bench_line_other_configurations(bench_line):
    """the other single-GPU BASELINE configurations and the reference's -n 1000 figure, in the same line"""
    d = bench_line
    o = d["others"]
    for k in ("config3", "config4"):
        assert 0.01 < o[k]["ms_per_frame"] < 1.0 and 0.1 < o[k]["frame_frac"] < 1.5 and "row_c2r_sharpen" in o[k]["kernel_ms"]
        assert abs(o[k]["frames_per_s"] - 1e3 / o[k]["ms_per_frame"]) < 1e-6 * o[k]["frames_per_s"]
    assert "-p 2" in o["config3"]["workload"] and "1920x1080" in o["config4"]["workload"]
    for k, bmin in (("config3", 3.0 * (2048 * 1024 + 4096 * 2048 * 2)), ("config4", 3.0 * (1920 * 1080 * 4 + 3840 * 2160 * 4)),
                    ("config3_u8_store", 3.0 * (2048 * 1024 + 4096 * 2048))):
        e = o[k]
        assert e["B_min"] == bmin and abs(e["b_min_frac"] - bmin / (e["ms_per_frame"] * 1e-3) / 8e12) < 1e-9 and e["b_min_frac"] < e["frame_frac"]
        assert e["energy_mj_per_frame"] is None or 10 < e["energy_mj_per_frame"] < 500
    # (outside BASELINE's list: the reference's third precision on the size-generic kernels, a plan-time size at -u 4/3)
    assert o["fp64"]["plan"].startswith("size-generic") and o["fhd_to_qhd_u4_3"]["plan"].startswith("specialised at plan time: u4/3")
    n = o["execute_n1000"]
    for k in ("config2", "config3", "config4"):
        # ms_per_iter: iterations in order on one stream (the reference's definition, = sequential_*); overlapped_*: the extension
        assert 0.01 < n[k]["ms_per_iter"] < 1.0 and n[k]["sequential_ms_per_iter"] == n[k]["ms_per_iter"] and 0.01 < n[k]["overlapped_ms_per_iter"] < 1.0
        assert set(n[k]["kernel_ms"]) >= {"row_r2c", "col_fwd_pad_inv", "row_c2r_sharpen"}
    # ... and the same figures as top-level keys of the line: the driver's parser keeps those
    assert d["execute_n1000"]["config2"]["ms_per_iter"] == n["config2"]["ms_per_iter"] and set(d["execute_n1000"]) == {"config2", "config3", "config4"}
    assert d["sequential_ms_per_iter"] == {k: n[k]["ms_per_iter"] for k in ("config2", "config3", "config4")}
    assert d["config3_frame_frac"] == o["config3"]["frame_frac"] and d["config4_frame_frac"] == o["config4"]["frame_frac"]


def test_bench_line_timing_relations(bench_line):
    """relations between the line's timings that hold on an undisturbed MI355X -- a slower or shared board reports, it does not abort"""
    d = bench_line
    o, n = d["others"], d["others"]["execute_n1000"]
    _soft(d["kernel_ms"]["-"] < 0.01, "empty kernel slot reads %.4f ms: event overhead not netted out" % d["kernel_ms"]["-"])
    for k in ("config2", "config3", "config4"):
        _soft(n[k]["overlapped_ms_per_iter"] <= 1.02 * n[k]["ms_per_iter"], "%s: overlapped iterations slower than ordered ones" % k)
        _soft(0.3 < n[k]["frame_frac"] <= n[k]["overlapped_frame_frac"] * 1.02 < 1.3, "%s: fractions out of range" % k)
    _soft(n["config2"]["overlapped_ms_per_iter"] <= 1.12 * d["ms_per_frame"], "overlapped iterations more than 12 % off the ring figure")
    _soft(n["config4"]["overlapped_ms_per_iter"] <= 1.12 * o["config4"]["ms_per_frame"], "1080p: overlapped iterations more than 12 % off the ring figure")
    _soft(n["config2"]["ms_per_iter"] >= 0.9 * d["ms_per_frame"], "ordered iterations faster than overlapped frames?")
    _soft(0.15 < o["fp64"]["ms_per_frame"] < 0.5 and 0.03 < o["fhd_to_qhd_u4_3"]["ms_per_frame"] < 0.12, "fp64 / u4/3 frame times out of their usual range")
    _soft(d["b_min_frac"] < (d["real_traffic_frac"] or 1.0) <= d["frame_roofline_frac"] * 1.02, "b_min < real traffic < B_alg fractions")


def test_bench_line_live_counters(bench_line):
    """HBM bytes measured by this very run (two rocprofv3 --pmc passes per configuration): needs rocprofv3 to work on the box"""
    d = bench_line
    ro, o = d["roofline"], d["others"]
    _soft(ro["traffic_source"].startswith("live: rocprofv3 --pmc"), "no live counters: " + ro["traffic_source"][-200:])
    assert 5e7 < ro["traffic"] < 4e8
    assert ro["traffic_live"]["row_c2r_sharpen"]["hbm_bytes_per_launch"] == ro["traffic"]
    for k, bmin in (("config3", 3.0 * (2048 * 1024 + 4096 * 2048 * 2)), ("config4", 3.0 * (1920 * 1080 * 4 + 3840 * 2160 * 4)),
                    ("config3_u8_store", 3.0 * (2048 * 1024 + 4096 * 2048))):
        e = o[k]
        _soft(e["frame_hbm_bytes_measured"] is not None and e["traffic_source"].startswith("live: rocprofv3 --pmc"), "%s: no live counters" % k)
        assert bmin <= e["frame_hbm_bytes_measured"] < 4e8 and e["b_min_frac"] <= e["real_traffic_frac"] < 1.0
        assert set(e["kernel_hbm_bytes_measured"]) == {"row_r2c", "col_fwd_pad_inv", "row_c2r_sharpen"}
    # (the 8-bit image is written once: 25 MB of writes -- round 3 wrote 76 MB --, the spectra, and the L2's reads of the lines it
    # merges the three planes' bytes into: ~84 MB of reads under the fused kernel's FETCH correction of round 6 (x1.44; 64 MB x1.0 before))
    assert o["config3_u8_store"]["kernel_hbm_bytes_measured"]["row_c2r_sharpen"] < 1.3e8
    assert o["config3_u8_store"]["kernel_hbm_bytes_measured"]["row_c2r_sharpen"] - 1.44 * 64e6 < 3.2e7      # ... i.e. the writes stay one image


def test_bench_line_committed_profiles_are_current(bench_line):
    """the static counter / trace figures were measured on the kernel sources that just ran (tools/gpu_round_end.sh refreshes them):
    a kernel edit without a refresh shows here, and nowhere else"""
    d = bench_line
    ro = d["roofline"]
    _soft(ro["traffic_kernel_sources_current"] is True, "profiles/hbm_traffic.json was measured on other kernel sources: re-run the PMC passes (tools/gpu_pmc.sh)")
    _soft(ro["traffic_static"] is not None and abs(ro["traffic"] - ro["traffic_static"]) < 0.1 * ro["traffic_static"], "live and committed HBM bytes differ by more than 10 %")
    if "frac_rocprof" in ro:
        _soft(ro["rocprof_kernel_sources_current"] is True, "profiles/kernel_stats_index.json was measured on other kernel sources")
        _soft(abs(ro["frac_rocprof"] - ro["frac"]) < 0.08, "HIP-event and rocprofv3 kernel durations disagree: %.3f vs %.3f" % (ro["frac"], ro["frac_rocprof"]))
    # what the frame is NOT bound by: vector-ALU issue slots in use (committed SQ_INSTS_VALU x 4 cycles over SIMDs x clock x time)
    _soft(d["frame_valu_insts"] is not None and 1.5e7 < d["frame_valu_insts"] < 3e7 and (d["valu_busy_frac"] is None or 0.3 < d["valu_busy_frac"] < 1.0),
          "committed vector-instruction counts missing or out of range")


def test_bench_line_rccl_selfcheck(bench_line):
    """every N = 1 line brings RCCL up as a one-rank communicator on this GPU (the library the N > 1 runs depend on)"""
    chk = bench_line["rccl_selfcheck"]
    _soft(chk.get("ok") is True and chk.get("ranks") == 1, "RCCL did not come up in the bench line: %s" % chk)


@pytest.mark.parametrize("png", [False, True])
def test_bench_host_streamed_lines(png):
    """--host-streamed (PCIe-inclusive, never the headline): pixels back, or -- --png -- finished PNG files encoded on the device;
    the line says which, and how many bytes crossed the link per frame"""
    r = subprocess.run([sys.executable, "bench.py", "--host-streamed", "--steps", "2", "--warmup", "1", "--repeats", "1", "--frames-per-step", "16",
                        "--ring", "4", "--width", "512", "--height", "256", "--no-cpu-baseline"] + (["--png"] if png else []),
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["pcie_inclusive"] is True and d["value"] > 100 and "HOST-STREAMED" in d["config"]["workload"]
    pixels = 3.0 * (512 * 256 + 1024 * 512)
    if png:
        assert "PNG" in d["config"]["workload"] and 0.2 * 3 * 1024 * 512 < d["png_bytes_per_frame"] < 3 * 1024 * 512
        assert abs(d["pcie_bytes_per_frame"] - (3 * 512 * 256 + d["png_bytes_per_frame"])) < 1
    else:
        assert d["pcie_bytes_per_frame"] == pixels and "png_bytes_per_frame" not in d


def test_bench_two_ranks_config5_over_gloo():
    env = dict(os.environ, FFTUP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--repeats", "2",
                        "--preset", "config5", "--profile-iters", "3"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["preset"] == "config5" and d["config"]["frames_per_step"] == 64
    assert d["dtype"] == "f16-memory/f32-math" and "uint8 RGB (fused load)" in d["config"]["workload"]
    assert abs(d["value"] - 2 * 2 * 64 / d["timed_region_s_median"]) < 1e-6 * d["value"]
    assert "cpu_baseline" not in d and "others" not in d     # rank 0 at N = 1 only
    # job accounting: the collective really had two ranks, every frame of the 128-frame job was processed once
    j = d["job"]
    assert d["rccl_ranks"] == 2 and j["collective_ranks"] == 2 and j["backend"] == "gloo"
    assert d["frames_done"] == 128 and j["frames_per_step_total"] == 128
    assert [r["rank"] for r in j["ranks"]] == [0, 1] and [r["first_frames"] for r in j["ranks"]] == [[0, 2, 4], [1, 3, 5]]
    assert all(len(r["pci_bus_id"]) >= 7 for r in j["ranks"]) and j["distinct_devices"] == 1      # (two ranks share this box's one GPU)
    assert d["checksum"] > 0


def test_bench_two_ranks_default_preset_says_what_ran():
    """the driver's scaling runs use the default preset: such a line, too, carries the ranks the collective really had and one
    device (PCI bus id) per rank"""
    env = dict(os.environ, FFTUP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29521", "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--repeats", "1",
                        "--frames-per-step", "32", "--profile-iters", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["preset"] == "config2" and d["rccl_ranks"] == 2 and d["frames_done"] == 2 * 32
    # every rank's own rate (frames / time until ITS work was done): a sagging rank is visible, the aggregate is below their sum only by the barrier
    assert len(d["per_rank_frames_per_s"]) == 2 and all(x > 100 for x in d["per_rank_frames_per_s"]) and d["value"] <= 1.001 * sum(d["per_rank_frames_per_s"])
    j = d["job"]
    assert [x["rank"] for x in j["ranks"]] == [0, 1] and [x["frames"] for x in j["ranks"]] == [[0, 7], [8, 15]]
    assert j["distinct_resident_frames"] == 16 and j["checksum"] > 0 and j["ranks"][0]["checksum"] != j["ranks"][1]["checksum"]


_job_cache = {}


def _job(nproc, frames_per_step, port):
    if (nproc, frames_per_step) in _job_cache:              # (the one-rank reference of 16 frames serves two tests)
        return _job_cache[(nproc, frames_per_step)]
    _job_cache[(nproc, frames_per_step)] = d = _job_run(nproc, frames_per_step, port)
    return d


def _job_run(nproc, frames_per_step, port):
    env = dict(os.environ, FFTUP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["bench.py", "--gpus", str(nproc), "--steps", "1", "--warmup", "1", "--repeats", "1", "--precision", "2", "--fuse-u8", "--job",
            "--frames-per-step", str(frames_per_step), "--ring", str(frames_per_step), "--profile-iters", "2", "--no-cpu-baseline", "--no-others",
            "--width", "512", "--height", "256"]
    cmd = [sys.executable] + args if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port)] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r.stdout)


def test_job_checksum_is_independent_of_the_sharding():
    """The same 16-frame job on one rank and striped over two (frame f*2 + r on rank r, VkResample.cpp:1622-1629): same
    number of frames done, same checksum of the outputs -- each frame was processed exactly once, whoever did it."""
    one = _job(1, 16, 0)
    two = _job(2, 8, 29519)
    assert one["frames_done"] == two["frames_done"] == 16
    assert one["checksum"] == two["checksum"] and one["checksum"] > 0
    assert two["rccl_ranks"] == 2 and one["rccl_ranks"] == 1


def test_job_through_the_shared_counter_has_the_stripes_checksum():
    """--queue: the same 16-frame job with the ranks claiming chunks of 4 frames from one counter (shard.FrameQueue) -- whoever
    processed which frame, every frame was processed exactly once: frames_done and the checksum of the outputs equal the
    stripe's (and the single-process run's)."""
    env = dict(os.environ, FFTUP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    base = ["--steps", "2", "--warmup", "1", "--repeats", "1", "--precision", "2", "--fuse-u8", "--job", "--profile-iters", "2",
            "--no-cpu-baseline", "--no-others", "--width", "512", "--height", "256"]
    one = _job(1, 16, 0)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29523", "bench.py", "--gpus", "2", "--frames-per-step", "8", "--queue", "--queue-chunk", "4"] + base,
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    two = _line(r.stdout)
    assert two["frames_done"] == one["frames_done"] == 16 and two["checksum"] == one["checksum"] > 0
    assert two["rccl_ranks"] == 2 and "shared counter" in two["job"]["stripe"]
    assert sum(x["frames"] for x in two["job"]["ranks"]) == 16
    q1 = subprocess.run([sys.executable, "bench.py", "--frames-per-step", "16", "--queue", "--queue-chunk", "4"] + base,
                        cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert q1.returncode == 0, q1.stderr[-3000:]
    assert _line(q1.stdout)["checksum"] == one["checksum"]


# ---- N ranks started the way a user (or the driver) starts them --------------------------------------------
def test_bench_gpus_flag_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no torch.distributed.run around it starts the two ranks itself (the reference fans its
    threads out inside the process, VkResample.cpp:1959-1969): the line says n_gpus = 2 and the collective had two ranks.
    (gloo dry run: the two ranks share this box's one GPU)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FFTUP_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--repeats", "1", "--frames-per-step", "32",
                        "--profile-iters", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["frames_done"] == 64 and d["job"]["backend"] == "gloo"
    assert abs(d["value"] - 2 * 32 / d["timed_region_s_median"]) < 1e-6 * d["value"]


def test_bench_refuses_rccl_ranks_without_a_gpu_each():
    """without the dry-run override the ranks talk RCCL, one per GPU: two ranks on a one-GPU box is an error, not a line"""
    import vkresample_amd as v
    if v.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    env = {k: v_ for k, v_ in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FFTUP_BENCH_BACKEND")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0", "--repeats", "1", "--frames-per-step", "8"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "visible GPU" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rccl_comes_up_on_this_gpu():
    """RCCL (torch.distributed backend "nccl") initialises on the MI355X as a one-rank communicator and reduces: what every
    N = 1 bench line carries as `rccl_selfcheck`"""
    sys.path.insert(0, ROOT)
    import bench
    chk = bench.rccl_selfcheck(0)
    print("RCCL selfcheck:", chk)
    assert chk["ok"] is True and chk["ranks"] == 1 and chk["version"], chk


def _config5(nproc, port, extra, frames_per_step=64):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FFTUP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["bench.py", "--gpus", str(nproc), "--steps", "1", "--warmup", "1", "--repeats", "1", "--precision", "2", "--fuse-u8", "--job",
            "--frames-per-step", str(frames_per_step), "--ring", str(frames_per_step), "--profile-iters", "2", "--no-cpu-baseline",
            "--no-others", "--no-rccl-check"] + extra
    r = subprocess.run([sys.executable] + args, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return _line(r.stdout)


def test_eight_rank_dry_run_of_config5():
    """BASELINE config 5 -- 512 synthetic 2048x1024 frames, -u 2 -p 2, over 8 ranks -- as a dry run on this box's one GPU (gloo):
    the 8-rank stripe (frame f*8 + r on rank r, 64 resident frames per rank) processes each of the 512 frames once and ends with
    the checksum ONE rank gets for the same 512 frames; so does the shared counter (--queue) on a job of 128 frames."""
    one = _config5(1, 0, [], frames_per_step=512)
    assert one["frames_done"] == 512 and one["rccl_ranks"] == 1 and one["checksum"] > 0
    eight = _config5(8, 0, ["--preset", "config5"])
    assert eight["n_gpus"] == 8 and eight["rccl_ranks"] == 8 and eight["job"]["collective_ranks"] == 8
    assert eight["frames_done"] == 512 and eight["job"]["frames_per_step_total"] == 512 and eight["checksum"] == one["checksum"]
    assert [x["rank"] for x in eight["job"]["ranks"]] == list(range(8))
    assert [x["first_frames"] for x in eight["job"]["ranks"]] == [[r, r + 8, r + 16] for r in range(8)]
    assert abs(eight["value"] - 8 * 64 / eight["timed_region_s_median"]) < 1e-6 * eight["value"]
    if os.environ.get("FFTUP_BIG_TESTS", "0") == "0":
        return                                               # (the shared counter over 8 ranks: FFTUP_BIG_TESTS=1; over 2 ranks it runs in test_job_through_the_shared_counter..)
    # the shared counter over 8 ranks (every rank keeps the step's 128 frames resident)
    q1 = _config5(1, 0, [], frames_per_step=128)
    q8 = _config5(8, 0, ["--queue", "--queue-chunk", "4"], frames_per_step=16)
    assert q8["rccl_ranks"] == 8 and q8["frames_done"] == q1["frames_done"] == 128 and q8["checksum"] == q1["checksum"]
    assert sum(x["frames"] for x in q8["job"]["ranks"]) == 128 and "shared counter" in q8["job"]["stripe"]
