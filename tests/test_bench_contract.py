"""bench.py prints ONE JSON line with the fields the driver reads (checked on a tiny workload)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--spp", "8", "--no-cpu"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["unit"] == "Mpaths/s" and out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    assert "cpu_baseline" not in out  # --no-cpu
    assert out["n_ranks_seen"] == 1 and "no reduce at N=1" in out["config"]["partition"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["c1", "c3", "c5"])
def test_every_config_prints_a_roofline_fraction_below_one(config):
    """BASELINE configs 1, 3 and 5 at a tiny sample count: the line carries a roofline fraction in (0, 1) -- for config 5 the
    counters' HBM-side bytes, with the walks' algorithmic bytes as a side field -- and the committed profile's counters."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", config, "--steps", "1", "--warmup", "1", "--spp", "4",
                        "--no-cpu"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    r = out["roofline"]
    assert 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4, r
    assert r["bound"] == ("hbm" if config == "c5" else "valu_fp32")
    c = r["counters"]
    for key in ("valu_issue_frac", "lane_util", "wait_frac", "gpu_rays_per_path"):
        assert 0 < c[key], (key, c)
    assert c["valu_issue_frac"] < 1 and c["lane_util"] <= 1 and c["wait_frac"] < 1
    if config == "c5":
        assert r["algorithmic"]["bytes_per_path"] > r["per_path"] and "frac" not in r["gather"]
    assert "other_configs" not in out


@pytest.mark.gpu
def test_default_line_carries_configs_1_3_5():
    """The driver's command (no --config): after the headline's steps the same process times configs 1, 3 and 5 and attaches them
    as other_configs; the headline's own fields are the contract's."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["metric"].startswith("Mpaths/sec (node), Cornell box 1024x1024 1024spp") and 0 < out["roofline"]["frac"] < 1
    assert out["roofline"]["counters"]["gpu_rays_per_path"] > 1
    others = out["other_configs"]
    assert [o["workload"].split()[0] for o in others] == ["random_spheres", "next_week_final", "sphere_field_1m"]
    for o in others:
        assert o["value"] > 0 and o["kernel_ms"] > 0 and 0 < o["roofline"]["frac"] < 1, o


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` without torch.distributed.run around it: the parent must start the two ranks itself
    (a child process, before it touches torch or HIP), relay rank 0's single JSON line and return the child's exit code.
    No GPU here: RT_BENCH_TEST_HOOK swaps the renderer for the CPU oracle and the backend for gloo, everything else --
    launch, sample ranges, reduce, report -- is the code the driver runs."""
    env = dict(os.environ, RT_BENCH_TEST_HOOK="bench_hook:oracle_renderer")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "24",
                        "--spp", "6"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path), env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 1
    assert "one RCCL sum-reduce" in out["config"]["partition"] and "2 GPUs" in out["config"]["partition"]
    assert out["value"] > 0 and "cpu_baseline" not in out
    # a failing child's exit code comes back
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "24",
                          "--spp", "6"], capture_output=True, text=True, timeout=600, cwd=str(tmp_path),
                         env=dict(env, RT_BENCH_TEST_HOOK="bench_hook:no_such_function"))
    assert bad.returncode != 0 and not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
