"""bench.py's own multi-rank path on CPU: `python bench.py --gpus 2` must START two ranks (fresh child processes, the driver's
launcher), run its rank / barrier / MAX-over-ranks logic over gloo and print ONE JSON line with n_gpus == 2 (round-2 verdict: the
flag used to be dead code).  The stub workload (5 ms of host time, 16 patches per step) stands in for the GPU step."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = dict(os.environ, KMSR_BENCH_BACKEND="gloo", **env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--no-cpu-baseline"] + args,
                          capture_output=True, text=True, env=e, timeout=300)


def _line(proc):
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout           # exactly one line on stdout: rank 0's JSON
    return json.loads(lines[0])


def test_bench_gpus_2_launches_two_ranks():
    one = _line(_run(["--gpus", "1", "--steps", "20", "--warmup", "2"]))
    two = _line(_run(["--gpus", "2", "--steps", "20", "--warmup", "2"]))
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["steps"] == 20 and two["warmup"] == 2
    assert one["ranks_seen"] == 1 and two["ranks_seen"] == 2 and two["backend"] == "gloo"      # an all_gather heard from 2 distinct ranks
    assert two["scaling"] == "weak" and two["config"]["workload"] == "stub_sleep5ms_b16"
    for r in (one, two):        # value = patches of ALL ranks / max-over-ranks wall time
        assert abs(r["value"] - 16 * r["n_gpus"] * 1e3 / r["ms_per_step"]) < 1e-2 * r["value"]
        assert 5.0 <= r["ms_per_step"] < 50.0
    # weak scaling of a fixed-duration step: two ranks ~ twice the per-rank rate (loose: host timer jitter)
    assert 1.4 < two["value"] / one["value"] < 2.6


def test_bench_launcher_reports_a_failed_rank():
    proc = _run(["--gpus", "2", "--steps", "3", "--warmup", "1"], KMSR_BENCH_STUB_FAIL_RANK="1")
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]     # no result line from a broken job


def test_bench_refuses_gpus_world_size_mismatch():
    """Under an external launcher whose WORLD_SIZE differs from --gpus the run must not print a line (round-3 advisor finding:
    it used to warn and go on with n_gpus = the launcher's value)."""
    e = dict(os.environ, KMSR_BENCH_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stub", "--no-cpu-baseline", "--gpus", "2",
                           "--steps", "2", "--warmup", "1"], capture_output=True, text=True, env=e, timeout=120)
    assert proc.returncode != 0 and "WORLD_SIZE=1" in proc.stderr and not proc.stdout.strip()


def test_bench_secondary_watchdog_still_reports_the_headline():
    """A secondary leg that never returns on N > 1 ranks (a stalled collective) must not cost the measured headline: every rank
    arms the same deadline, rank 0 prints the line it already holds with secondary marked failed, all ranks exit 0."""
    ok = _line(_run(["--gpus", "2", "--steps", "3", "--warmup", "1"], KMSR_BENCH_STUB_SECONDARY="ok"))
    assert ok["secondary"] == {"stub": True} and ok["n_gpus"] == 2
    for mode in ("hang", "hang1"):
        r = _line(_run(["--gpus", "2", "--steps", "3", "--warmup", "1"], KMSR_BENCH_STUB_SECONDARY=mode, KMSR_BENCH_SECONDARY_TIMEOUT="3"))
        assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["value"] > 0
        if mode == "hang":
            assert r["secondary"]["failed"] and "watchdog" in r["secondary"]["error"]
        else:                      # rank 0's own legs finished; rank 1 left through its watchdog
            assert r["secondary"] == {"stub": True}
    # a single rank arms no watchdog (nothing to wait for)
    one = _line(_run(["--gpus", "1", "--steps", "3", "--warmup", "1"], KMSR_BENCH_STUB_SECONDARY="ok"))
    assert one["secondary"] == {"stub": True}
