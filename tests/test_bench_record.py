"""The measurement record of a run is the LAST line of its output that holds a JSON object (the driver reads the combined
stdout + stderr tail).  Round 5's record did not parse: a brace-bearing summary on stderr came after the stdout line, and
the line itself had grown to 20 KB.  These CPU tests emulate that reading of the output, with braces deliberately written
to both streams around the line, and bound the size of the line bench.py composes from a full set of results."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SECONDARY = ("other_configs", "sfno_config5", "sfno_w16_gelu", "sfno_w20", "sfno_w20_gelu", "sfno_w32", "sfno_notebook_training")


def read_like_the_driver(stdout: str, stderr: str, tail_bytes: int = 8192):
    """The last line holding a brace in the last `tail_bytes` of the output of each stream, stderr after stdout."""
    text = stdout[-tail_bytes:] + "\n---- stderr ----\n" + stderr[-tail_bytes:]
    lines = [ln for ln in text.splitlines() if "{" in ln]
    assert lines, text[-500:]
    return json.loads(lines[-1])


def _run(args, noise=True):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    if noise:
        env["BENCH_TEST_STDERR_NOISE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stdout.decode(), r.stderr.decode()


@pytest.mark.parametrize("gpus", [1, 2])
def test_last_json_line_of_the_output_is_the_record(gpus):
    out, err = _run(["--gpus", str(gpus), "--steps", "3", "--host-only"])
    assert "noise" in err and "{" not in err and "}" not in err      # the noise arrived, its braces did not
    rec = read_like_the_driver(out, err)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "dtype", "roofline", "cpu_baseline"):
        assert key in rec, key
    assert rec["n_gpus"] == gpus and rec["steps"] == 3 and rec["ms_per_step"] > 0
    assert [ln for ln in out.splitlines() if ln.strip()] == [json.dumps(rec)] or len(out.splitlines()) == 1


def test_composed_line_is_small_and_complete():
    """bench.compose_line on a full set of results (round 5's 20 KB record, committed under profiles/) gives a line under
    8 KB that holds every key the record needs, and a detail record that loses nothing."""
    sys.path.insert(0, ROOT)
    import bench

    full = json.load(open(os.path.join(ROOT, "profiles", "r05_final_bench.json")))
    out = {k: v for k, v in full.items() if k not in SECONDARY and k != "summary"}
    detail = {"secondary": {k: full[k] for k in SECONDARY if k in full}, "regions": [{"seconds": 0.14, "per_rank_seconds": [0.14]}] * 5}
    line, everything = bench.compose_line(out, detail)
    text = json.dumps(line)
    assert len(text.encode()) < 8192, len(text)
    rec = read_like_the_driver("x" * 20000 + "\n" + text + "\n", "/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n")
    assert rec["metric"] and rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["dtype"] == "f64"
    assert rec["config"]["workload"] and 0 < rec["roofline"]["frac"] < 1 and rec["roofline"]["bound"] == "hbm"
    assert rec["roofline"]["unit"] == "GB/s" and rec["roofline"]["peak"] == 8000.0 and rec["roofline"]["achieved"] > 0
    assert rec["cpu_baseline"]["value"] > 0 and rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] > 0
    assert rec["kernels"] and rec["hbm_probe"] and rec["roofline_worst"]["frac"] <= rec["roofline"]["frac"]
    assert list(rec)[-1] == "summary" and rec["summary"]["steps_per_s"] == rec["value"]
    # at most a handful of scalars per secondary workload in the line, everything in the detail record
    for key in SECONDARY:
        assert key in rec and len(json.dumps(rec[key])) < 1200, key
        assert everything["secondary"][key] == full[key]
    assert rec["sfno_config5"]["forward_plus_loss_ms"] == full["sfno_config5"]["forward_plus_loss_ms"]
    assert rec["other_configs"]["C2_256x16_f32"]["steps_per_s"] == full["other_configs"]["C2_256x16_f32"]["steps_per_s"]
    # a secondary job that died leaves the headline untouched
    line2, _ = bench.compose_line(out, {"secondary": {"error": "bench_secondary.py did not finish within 420 s"}})
    assert line2["value"] == line["value"] and "secondary_error" in line2 and len(json.dumps(line2)) < 8192


def test_nothing_in_bench_writes_braces_after_the_line():
    """No write to stderr or stdout in bench.py other than the one os.write of the line (static check of the source)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "bench summary" not in src.replace('bench summary: {"noise"', "")
    assert src.count("os.write(real_stdout") == 3      # the host-only line, the --c4-only object, the line
