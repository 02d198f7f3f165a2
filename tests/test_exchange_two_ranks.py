"""The library's exchange steps (csrc/comm.h: kmg_match_sharded, kmg_db_exchange_by_key_range) with MORE THAN ONE RANK.

RCCL refuses two ranks on one device and the GPU boxes here have one, so two fresh processes on GPU 0 -- each with its own
kmg_ctx -- talk through a test-only transport double (tests/transport_double/rccl_double.cpp: the nine symbols comm.h
looks up by name, host staged through shared memory, strict about mismatched sends and receives).  What runs is the
library's own code for N ranks: the all-gather of the count matrix with its status column, the grouped send / receive
loops, the one-launch join of the received runs, the reverse exchange, the rebuilding and merging of received database
parts -- every rank's results against the CPU oracle (tests/_exchange_rank.py).  Where two real GPUs exist the same
script runs on real RCCL (test_exchange_two_gpus_real_rccl)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE_DIR = os.path.join(ROOT, "tests", "transport_double")


def _run_ranks(nranks, tmp_path, real=False, k=21):
    env = dict(os.environ, KMG_TEST_REAL_RCCL="1" if real else "0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(ROOT, "tests", "_exchange_rank.py"), str(r), str(nranks), str(tmp_path), str(k)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT, env=env) for r in range(nranks)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            out, _ = p.communicate()
            outs.append(out.decode(errors="replace") + "\n[timed out]")
            continue
        outs.append(out.decode(errors="replace"))
    for r, p in enumerate(procs):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, outs[r][-4000:])
    return [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(nranks)]


@pytest.mark.gpu
@pytest.mark.parametrize("nranks,k", [(2, 21), (3, 21), (2, 40), (3, 64)])
def test_exchange_with_several_ranks_on_one_gpu(nranks, k, tmp_path):
    assert os.path.exists(os.path.join(DOUBLE_DIR, "librccl.so.1")), "run __graft_entry__.build() (make -C tests/transport_double)"
    res = _run_ranks(nranks, tmp_path, k=k)          # (k > 32: two words a k-mer)
    for r in res:
        assert r["ok"] and r["cases"] == 9
        # the grouped exchange and the count all-gather really went through the double, with peers on the other side
        assert r["allgathers"] >= 9 and r["rounds"] > r["allgathers"]
    assert sum(r["sends"] for r in res) == sum(r["receives"] for r in res) > 0


@pytest.mark.gpu
def test_exchange_two_gpus_real_rccl(tmp_path):
    """The same ranks on RCCL itself, one GPU each -- wherever two GPUs are visible (the one-GPU boxes skip it)."""
    import torch
    if torch.cuda.device_count() < 2:          # (counting devices does not initialise the GPU)
        pytest.skip("needs two GPUs")
    for r in _run_ranks(2, tmp_path, real=True):
        assert r["ok"] and r["cases"] == 9
