"""bench.py's launch contract, the part that needs no GPU: --gpus must match WORLD_SIZE (a mismatch exits non-zero before anything is
imported or initialised -- never a silent 1-GPU run that prints n_gpus: 1)."""
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
