"""bench.py's launch contract, the part that needs no GPU: --gpus must match WORLD_SIZE (a mismatch exits non-zero before anything is
imported or initialised -- never a silent 1-GPU run that prints n_gpus: 1)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_must_match_world_size():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode != 0
    assert "WORLD_SIZE=2" in out.stderr and not out.stdout.strip()


def test_minimal_depth_is_the_reference_workaround():
    sys.path.insert(0, ROOT)
    import bench
    assert [bench.minimal_depth(n) for n in (100000, 1000000, 4000000)] == [2, 5, 7]


def _run_supervisor(tmp_path, child_src, grace="3"):
    child = tmp_path / "child.py"
    child.write_text(child_src)
    env = dict(os.environ, HMX_BENCH_CHILD_SCRIPT=str(child), HMX_BENCH_RANK0_GRACE=grace)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)


def test_supervisor_relays_the_measurement_completed_before_a_rank_died(tmp_path):
    """a rank that aborts after the plain exchange was timed must not cost that measurement: rank 0 left it in the side file."""
    r = _run_supervisor(tmp_path, "import os, sys, time, json\n"
                        "if os.environ['RANK'] == '0':\n"
                        "    open(os.environ['HMX_BENCH_SIDE_FILE'], 'w').write(json.dumps({'metric': 'm', 'value': 1.5}) + '\\n')\n"
                        "    time.sleep(60)\n"
                        "else:\n"
                        "    time.sleep(1.0)\n"
                        "    sys.exit(3)\n", grace="1")
    assert r.returncode == 1
    assert json.loads(r.stdout.strip().splitlines()[-1]) == {"metric": "m", "value": 1.5}
    assert "relaying" in r.stderr


def test_supervisor_prefers_rank0s_own_line_and_reports_nothing_when_there_is_none(tmp_path):
    r = _run_supervisor(tmp_path, "import os, sys, time\n"
                        "if os.environ['RANK'] == '0':\n"
                        "    print('{\"metric\": \"own\"}', flush=True)\n"
                        "    time.sleep(60)\n"
                        "else:\n"
                        "    time.sleep(1.0)\n"
                        "    sys.exit(3)\n", grace="1")
    assert r.returncode == 1 and json.loads(r.stdout.strip()) == {"metric": "own"}
    r = _run_supervisor(tmp_path, "import os, sys, time\n"
                        "time.sleep(0.5 if os.environ['RANK'] == '1' else 60)\n"
                        "sys.exit(3)\n", grace="1")
    assert r.returncode == 1 and r.stdout.strip() == ""
