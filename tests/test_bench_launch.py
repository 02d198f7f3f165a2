"""bench.py's own launcher (`python bench.py --gpus N` without torchrun): N fresh rank processes with the launch
contract's environment, started before the parent touches a GPU.  CPU tests: the spawn path drives a world-2 gloo
run of the N > 1 line's checked step (tests/_rank_probe.py), a dying rank takes the launch down with a nonzero
status, and asking for more GPUs than the box has fails loudly instead of printing a one-GPU line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tests", "_rank_probe.py")


def test_launcher_runs_two_ranks_through_the_checked_step(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    rc = bench.launch_ranks(2, script=PROBE, argv=[str(tmp_path)], need_gpus=False)
    assert rc == 0
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert [g["rank"] for g in got] == [0, 1] and all(g["world"] == 2 for g in got)
    assert got[0]["hits_key_range"] == got[0]["hits_whole"] == got[1]["hits_key_range"] > 0


def test_launcher_reports_a_dead_rank(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    assert bench.launch_ranks(2, script=PROBE, argv=[str(tmp_path), "fail"], need_gpus=False, grace=3) != 0


def test_more_gpus_than_the_box_has_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two GPUs: the launch would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and p.stdout.strip() == ""
    assert "--gpus 2 asked for" in p.stderr


def test_balanced_ranges_may_be_empty_and_still_tile_the_key_space():
    from kmers_amd import dist as kd
    hist = np.zeros(64, np.int64)
    hist[0] = 1000                                           # poly-A: everything in bucket 0, world of 3
    r = kd.balanced_key_ranges(hist, 3)
    assert r[0][0] == 0 and r[-1][1] == 64 and all(a[1] == b[0] for a, b in zip(r[:-1], r[1:]))
    assert all(a <= b for a, b in r) and any(a == b for a, b in r)
    # the search k-mers of an empty range are an empty slice
    keys = (np.arange(0, 64, dtype=np.uint64) << np.uint64(58))
    c = kd.cut_points(keys, r, 3)
    assert c[0] == 0 and c[-1] == 64 and np.all(np.diff(c) >= 0)
