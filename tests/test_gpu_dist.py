"""GPU (one device): the multi-rank path of bench.py rehearsed with two ranks sharing the card.

The 8-GPU run over RCCL is the driver's; what can be checked on a one-GPU box is everything around the
collective: `python bench.py --gpus 2` launching its own ranks, each rank running the HIP forward on
its shard with two lanes / streams in flight, the all-gather ordered behind the forward on the lane's
stream (gloo stands in for RCCL: TTNET_DIST_BACKEND), max-over-ranks timing, and -- with
--verify-gather -- the gathered logits being bit-identical to the logits one process computes for
all shards."""
import json
import os
import subprocess
import sys

import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_self_launched_gather_equals_single_process():
    env = dict(os.environ, TTNET_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
                        "--batch", "24", "--verify-gather", "--no-cpu-baseline"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 48 and d["scaling"] == "weak"
    assert d["gather_verified"] is True and d["dist_backend"] == "gloo"
    assert d["value"] > 0 and d["inflight"] == 2
