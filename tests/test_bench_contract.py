"""bench.py without a GPU: the module imports, its expectations come from the oracle fixture (28 levels, 319 228 361 states), the flags of
the driver's contract parse, the CPU-baseline leg runs the oracle and reports the cores it used, and without a HIP device the bench fails
loudly instead of falling back to anything."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_expectations_come_from_the_oracle_fixture():
    sys.path.insert(0, ROOT)
    import bench
    E = bench.EXPECT
    assert (E["distinct"], E["depth"], len(E["levels"])) == (319228361, 28, 28) and E["fixture"].startswith("oracle_levels_config2")
    assert E["checksums"] and E["viol_fp"] == int("8f264756864213e1", 16)
    with open(os.path.join(ROOT, "BASELINE.json")) as f:
        base = json.load(f)
    assert "distinct states/sec" in base["metric"]


def test_flags_of_the_contract_parse():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--no-config3", "--no-cpu-baseline"):
        assert flag in r.stdout


def test_cpu_baseline_leg_and_loud_failure_without_a_gpu():
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(1.5)
    assert cb["kind"] == "port" and cb["unit"] == "distinct states/s" and cb["value"] > 1e4 and 1 <= cb["cores"] <= (os.cpu_count() or 1)
    assert "oracle/vsr_oracle_mt" in cb["sample"]
    import vsr_tlaplus_amd as vt
    if vt.load().vsrmc_device_count() == 0:                       # this container: the product path must refuse, not fall back
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-config3"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and not any(l.startswith("{") for l in r.stdout.splitlines()), r.stdout[-500:]
