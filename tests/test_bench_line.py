"""The line the driver parses (bench.py: compact / emit) on a REAL full record: profiles/r06_bench_detail.json is what
`python bench.py` produced on an MI355X at the end of round 6 (every leg present).  Round 5's single 20.6 KB line did not fit
the driver's 8 KB tail and left BENCH_r05.json.parsed = null; the compact line must stay below 6 KB, carry the contract's keys,
the dominant kernel's roofline and the CPU baseline as flat scalars, and must not lose the headline numbers of the record."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_compact_line_of_a_full_record():
    b = _bench()
    with open(os.path.join(ROOT, "profiles", "r06_bench_detail.json")) as f:
        full = json.loads(f.read())
    assert len(json.dumps(full)) > 15000            # the record that no longer fits a line
    c = b.compact(full)
    line = json.dumps(c)
    assert len(line) < b.COMPACT_LIMIT == 6144, len(line)
    assert json.loads(line) == c
    for k in CONTRACT:
        assert k in c and (c[k] == full[k] or k == "config"), k
    assert c["config"]["workload"].startswith("configs[1]") and c["config"]["world_size_seen"] == full["env"]["world_size_seen"]
    roof, cpu = c["roofline"], c["cpu_baseline"]
    assert all(not isinstance(v, (dict, list)) for v in roof.values()) and len(roof) <= 24
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "bolfi_iters_per_s", "cfg3_e2e_wall_s",
              "cfg5_ms_acquire", "cfg4_strong_value"):
        assert roof[k] == full["roofline"][k], k
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12 and roof["bound"] == "hbm"
    for k in ("value", "unit", "cores", "kind", "sample", "bolfi_value", "bolfi_cores"):
        assert k in cpu, k
    assert len(cpu["sample"]) <= 120 and all(not isinstance(v, (dict, list)) for v in cpu.values())
    assert all(isinstance(v, (int, float)) for v in c["more"].values())
    assert c["detail"] == "bench_detail.json"
