"""The measurement tools stay runnable: the microbenchmarks compile for gfx950 (CPU check), the per-rank slab cost tool runs
end to end on a small lattice (GPU)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", ["tile_pull_copy.hip", "march_pull_copy.hip", "march3d_pull_copy.hip", "tile_persist.hip", "tile_compact.hip"])
def test_microbenchmarks_compile(src, tmp_path):
    out = tmp_path / "a.out"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tools", "microbench", src), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert out.exists()


def test_live_traffic_passes_stand_down_without_side_effects(monkeypatch, tmp_path):
    """bench.py counts its HBM bytes with rocprofv3 passes over a child process; it must start none when it is itself being profiled
    (rocprofv3 exports ROCPROF_* to the profiled process), when rocprofv3 is missing, and after one pass of the run has failed"""
    sys.path.insert(0, ROOT)
    import bench
    started = []
    monkeypatch.setattr(subprocess, "Popen", lambda *a, **k: started.append(a) or (_ for _ in ()).throw(AssertionError("a pass was started")))
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", str(tmp_path))
    assert bench.under_rocprof()
    bench._LIVE["ok"] = True
    assert bench.live_pmc_traffic("rk3dq_fused<false", ["--steps", "1"]) is None and not started
    # one failure ends the live passes of the run
    assert bench._LIVE["ok"] is False
    monkeypatch.delenv("ROCPROF_OUTPUT_PATH")
    assert not bench.under_rocprof()
    assert bench.live_pmc_traffic("rk3dq_fused<false", ["--steps", "1"]) is None and not started
    # no rocprofv3 on the machine
    bench._LIVE["ok"] = True
    monkeypatch.setattr(shutil, "which", lambda name: None)
    monkeypatch.setattr(os.path, "exists", lambda path: False)
    assert bench.live_pmc_traffic("rk3dq_fused<false", ["--steps", "1"]) is None and not started
    bench._LIVE["ok"] = True


@pytest.mark.gpu
def test_slab_rank_cost_runs_on_a_small_lattice():
    env = dict(os.environ, LBMPM_K3_RELAX="SRT")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slab_rank_cost.py"), "64", "2", "4"], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("rank ")]
    assert len(lines) == 2 and "sum over ranks" in r.stdout


@pytest.mark.gpu
def test_pipelined_slab_bench_reports_a_bit_equal_run():
    """tools/slabbench_pipelined.py: k ranks of lbmpm_rk3d_step_slab on one GPU (one host thread per rank, exact exchange) -- the tool
    that showed round 3's face-message bug; its verdict line must say the run equals the single domain"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "slabbench_pipelined.py"), "96", "3", "4"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, LBMPM_K3_RELAX="MRT"), cwd=ROOT)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "k=3 pipelined step_slab" in r.stdout and "bit for bit: True" in r.stdout, r.stdout[-1500:]
