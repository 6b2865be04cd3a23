"""bench.py --gpus N started WITHOUT a launcher must start its own N ranks (fresh children, before any GPU call) and report the
number of ranks that really joined; a launcher whose WORLD_SIZE disagrees with --gpus is an error, not a silent 1-rank run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


def test_self_launch_two_ranks_joins_two():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"], env=_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["backend"] in ("gloo", "nccl")


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])


def test_hung_rank_is_killed_by_the_watchdog(tmp_path):
    """a rank that never comes back must not hang the caller: the wall-clock watchdog terminates the children and exits non-zero; every
    rank leaves a log file"""
    import time

    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TEST_HANG="1", ISEG_BENCH_TIMEOUT_S="25", ISEG_BENCH_LOG_DIR=str(tmp_path)), capture_output=True,
                       text=True, timeout=200)
    assert r.returncode == 4, (r.returncode, r.stderr[-1500:])
    assert time.time() - t0 < 120
    assert "terminating the remaining ranks" in r.stderr
    assert sorted(os.listdir(tmp_path)) == ["rank0.log", "rank1.log"]


def test_failed_rank_takes_the_job_down(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TEST_FAIL="1", ISEG_BENCH_TIMEOUT_S="120", ISEG_BENCH_LOG_DIR=str(tmp_path)), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 1, (r.returncode, r.stderr[-1500:])
    assert "(1, 7)" in r.stderr and "rank 1 (tail)" in r.stderr


def test_launcher_parent_counts_gpus_without_touching_them():
    """the parent decides from the visibility list / kfd topology, never from torch.cuda.* (which initialises the HIP runtime)"""
    import re

    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def self_launch"):src.index("def joined_ranks")]
    assert not re.search(r"torch\.cuda\.", body.split('"""', 2)[2])
    code = "import os, sys; sys.path.insert(0, %r); import bench; print(bench.visible_gpu_count())" % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=_env(HIP_VISIBLE_DEVICES="0,1,2"), capture_output=True, text=True, timeout=120)
    assert out.stdout.strip().splitlines()[-1] == "3", out.stderr[-500:]
    out = subprocess.run([sys.executable, "-c", code], env=_env(HIP_VISIBLE_DEVICES=""), capture_output=True, text=True, timeout=120)
    assert out.stdout.strip().splitlines()[-1] == "0"


def _probe_run(mode, tmp_path, timeout_s="20"):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TEST_PROBE=mode, ISEG_BENCH_PROBE_TIMEOUT_S=timeout_s, ISEG_BENCH_TIMEOUT_S="200",
                                ISEG_BENCH_LOG_DIR=str(tmp_path)), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), r


def test_exchange_probe_passes_on_every_rank_selects_the_stream_ordered_exchange(tmp_path):
    """bench.py --gpus N (round-4 verdict, item 5): every rank starts a fresh probe child before it touches the GPU; the children's common
    all-reduce(MIN) of success flags prints the marker on every rank -> the measured job runs the stream-ordered RCCL exchange (+ graph replay)"""
    out, _ = _probe_run("ok", tmp_path)
    assert out["n_gpus"] == 2 and out["exchange"].startswith("stream-ordered RCCL"), out


def test_exchange_probe_failure_on_one_rank_falls_back_on_every_rank(tmp_path):
    """rank 1's probe child dies before the agreement: rank 0's child never gets its all-reduce partner, is killed by pid at the probe's own
    timeout, and BOTH ranks take the c10d + eager branch (a mixed choice would hang the measured job); the job itself still completes"""
    out, r = _probe_run("fail", tmp_path)
    assert out["n_gpus"] == 2 and out["exchange"].startswith("c10d work objects"), out
    logs = "".join(open(os.path.join(tmp_path, f)).read() for f in sorted(os.listdir(tmp_path)))
    assert logs.count("native-exchange probe did not pass") == 2, logs[-1500:]


def test_exchange_probe_hang_is_cut_off_by_its_timeout(tmp_path):
    out, _ = _probe_run("hang", tmp_path, timeout_s="15")
    assert out["exchange"].startswith("c10d work objects") and "no result after 15 s" in out["exchange"], out


def test_an_explicit_exchange_choice_skips_the_probe():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TEST_PROBE="hang", ISEG_DIST_NATIVE="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "set by the caller" in out["exchange"]


def test_ranks_that_decided_differently_agree_on_the_fallback(tmp_path):
    """round-5 advisor: each rank decides alone from its own probe child; when the decisions differ (start-up skew around the probe's limit) the
    ranks meet in an all-reduce(MIN) of their decisions before the first collective of the job -- here rank 1's parent pretends it lost its probe
    while rank 0's passed: rank 0 must come down to the c10d exchange too"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TEST_PROBE="ok", ISEG_BENCH_TEST_SPLIT_DECISION="1", ISEG_BENCH_PROBE_TIMEOUT_S="60",
                                ISEG_BENCH_TIMEOUT_S="200", ISEG_BENCH_LOG_DIR=str(tmp_path)), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["exchange"].startswith("c10d work objects (another rank"), out
    log0 = open(os.path.join(tmp_path, "rank0.log")).read()
    assert "all ranks take c10d" in log0, log0[-800:]


def test_probe_port_comes_from_the_launcher_and_fp32_reaches_the_probe():
    """the launcher parent binds a free port for the probe's rendezvous (no MASTER_PORT + 23 guess); --fp32 is part of the probe's command line"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "ISEG_BENCH_PROBE_PORT" in src[src.index("def self_launch"):src.index("def joined_ranks")]
    body = src[src.index("def choose_exchange"):src.index("def probe_native")]
    assert 'cmd.append("--fp32")' in body and "probe_port()" in body
    code = ("import os, sys; sys.path.insert(0, %r); import bench; os.environ['ISEG_BENCH_PROBE_PORT'] = '43210'; print(bench.probe_port()); "
            "del os.environ['ISEG_BENCH_PROBE_PORT']; os.environ['MASTER_PORT'] = '30000'; print(bench.probe_port())" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert out.stdout.split()[-2:] == ["43210", "30023"], out.stderr[-500:]


def test_two_ranks_pin_themselves_to_disjoint_core_shares(tmp_path):
    """round-5 verdict item 6: every rank pins itself (affinity only, before any GPU call) to a share of its GPU's NUMA node -- without a kfd
    topology (this box) the allowed cores are split evenly"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check-launch"],
                       env=_env(ISEG_BENCH_TIMEOUT_S="200", ISEG_BENCH_LOG_DIR=str(tmp_path)), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    if len(os.sched_getaffinity(0)) >= 4:
        assert out["host_affinity"] and "cores" in out["host_affinity"], out
    code = "import os, sys; sys.path.insert(0, %r); import bench; print(bench.pin_to_gpu_numa(0, 1)); print(bench._cpulist('0-3,8,10-11'))" % ROOT
    o = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=120)
    assert o.stdout.strip().splitlines()[-2:] == ["None", "[0, 1, 2, 3, 8, 10, 11]"], o.stderr[-500:]
