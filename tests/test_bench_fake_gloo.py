"""bench.py's N > 1 control flow, executed: `python bench.py --gpus 2 --fake-device` (CPU tensors, gloo, the stand-in library
of bench_fake.py) must run the whole script -- self-launch under torch.distributed.run on 127.0.0.1, both scaling legs
(weak configs[1] with the per-step gather of the packed sampler states; strong configs[3] with the all-gather + fixed-order
merge of the column statistics), every post-run assertion of rank 0 -- and print ONE well-formed JSON line that says it is a
dry run.  No 8-GPU node has been available to the builder: this is what makes the first real multi-GPU run boring."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _run(gpus, extra=()):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--fake-device", "--steps", "5",
           "--warmup", "2"] + list(extra)
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    out = p.stdout.splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], "exactly one JSON line from rank 0, the last of stdout: got %d" % len(lines)
    # the driver keeps the last 8 KB of stdout: the line it parses must fit with room to spare (round 5's 20.6 KB line
    # left BENCH_r05.json.parsed = null)
    assert len(lines[0]) < 6144, len(lines[0])
    compact = json.loads(lines[0])
    assert json.loads(json.dumps(compact)) == compact
    detail = [ln for ln in p.stderr.splitlines() if ln.startswith("bench detail: ")]
    assert len(detail) == 1
    r = json.loads(detail[0][len("bench detail: "):])
    for k in REQUIRED:      # the compact line carries the contract's keys with the detail record's values
        assert compact[k] == r[k] or k in ("config", "roofline"), k
    assert compact["config"]["world_size_seen"] == gpus == r["env"]["world_size_seen"]
    assert compact["config"]["samples_per_gpu"] == r["config"]["samples_per_gpu"]
    assert "detail" not in compact["roofline"] and compact["roofline"]["frac"] == r["roofline"]["frac"]
    assert all(not isinstance(v, (dict, list)) for v in compact["roofline"].values())
    return r


def test_two_ranks_run_the_whole_script():
    r = _run(2)
    for k in REQUIRED:
        assert k in r, k
    assert r["n_gpus"] == 2 and r["env"]["world_size_seen"] == 2 and r["env"]["backend"] == "gloo"
    assert r["steps"] == 5 and r["warmup"] == 2 and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert r["data"].startswith("fake") and "NOT measurements" in r["config"]["dry_run"]
    # the weak leg gathered the packed state at the end of its (one, incomplete) exchange period of 8 steps; the nested strong
    # leg -- a whole SMC round per step -- gathers in every step
    assert r["config"]["exchange_every"] == 8 and r["env"]["gathers_per_rank"] >= 1 + 5
    every = _run(2, ["--exchange-every", "1"])
    assert every["config"]["exchange_every"] == 1 and every["env"]["gathers_per_rank"] >= 7 + 5
    assert r["value"] > 0 and abs(r["value"] - 2 * r["config"]["samples_per_gpu"] * 5 / (r["ms_per_step"] * 5e-3)) < 1e-6 * r["value"]
    c4 = r["cfg4_strong"]
    assert c4["scaling"] == "strong" and c4["n_gpus"] == 2 and c4["config"]["samples_per_gpu"] == 100000
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and sum(1 for v in roof.values() if not isinstance(v, (dict, list))) <= 24
    assert {"cfg4_strong_value", "cfg4_strong_kernel_frac"} <= set(roof) and "cfg4_strong_ms_per_step" in roof["detail"]


def test_one_rank_and_the_adaptive_workload():
    r = _run(1, ["--workload", "adaptive", "--scaling", "strong"])
    assert r["n_gpus"] == 1 and r["scaling"] == "strong" and r["data"].startswith("fake")
    assert r["config"]["samples_per_gpu"] == 200000
    r3 = _run(3, ["--workload", "adaptive", "--scaling", "strong", "--total", "100001"])
    assert r3["n_gpus"] == 3 and r3["env"]["world_size_seen"] == 3      # a total the ranks do not divide
