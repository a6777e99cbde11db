"""CPU: the C-ABI library loads and exports what include/ttnet.h declares (no compute
without a GPU), and the N>1 sharding / gather logic under gloo with world_size 2."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from _util import ROOT, golden_npz
from scale_imagenet_amd import _lib
from scale_imagenet_amd.dist import shard_bounds


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ttnet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ttnet_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from scale_imagenet_amd.build import build_lib
    build_lib(verbose=False)
    lib = C.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"libttnet.so does not export {name}"
    assert sorted(n for n, _, _ in _lib.SYMBOLS) == declared, "ctypes binding and header disagree"


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure():
    lib = _lib.load()
    desc = _lib.NetDesc(0, 8, 8, 1, 224, 224, 8, 0)
    h = C.c_void_p()
    st = lib.ttnet_plan_create(C.byref(desc), 0, C.byref(h))
    assert st < 0 and not h.value
    assert b"no HIP device" in lib.ttnet_last_error() or b"hip" in lib.ttnet_last_error().lower()
    with pytest.raises(_lib.TTNetError):
        _lib.check(st)


def test_shard_bounds():
    for n, w in [(4096, 8), (256, 2), (10, 4), (3, 8), (1, 1)]:
        spans = [shard_bounds(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert shard_bounds(4096, 3, 8) == (1536, 512)
    with pytest.raises(ValueError):
        shard_bounds(8, 8, 8)


_WORKER = r'''
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from scale_imagenet_amd import synth
from scale_imagenet_amd.dist import init_from_env, shard_bounds, all_gather_logits
from scale_imagenet_amd.spec import make_spec
from oracle import ttnet_float as OF
rank, world, _ = init_from_env("gloo")
n_total = {n_total}
spec = make_spec("xsmall")
sd = OF.to_torch_state(synth.synth_state_dict(spec))
first, count = shard_bounds(n_total, rank, world)
x = torch.from_numpy(synth.synth_images(count, first=first))
torch.set_num_threads(2)
local = OF.forward(x, sd, spec)           # the oracle stands in for the device forward here
allv = all_gather_logits(local, n_total)
if rank == 0:
    np.save({out!r}, allv.numpy())
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("n_total,launcher", [(4, "env"), (5, "spawn_ranks")])
def test_two_rank_gloo_shard_and_gather(tmp_path, n_total, launcher):
    """World size 2 over gloo: once with the rank environment written by hand (what torch.distributed.run
    provides), once through the package's own launcher (what `python bench.py --gpus 2` uses)."""
    out = str(tmp_path / "all.npy")
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, n_total=n_total, out=out))
    if launcher == "env":
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + n_total), WORLD_SIZE="2")
        procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)))
                 for r in range(2)]
        for p in procs:
            assert p.wait(timeout=300) == 0
    else:
        from scale_imagenet_amd.launch import spawn_ranks
        assert spawn_ranks([str(script)], 2, timeout_s=300) == 0
    got = np.load(out)
    ref = golden_npz("xsmall")["logits"][:n_total]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 1e-5          # same images, same order, whatever the split
    assert np.array_equal(got.argmax(1), ref.argmax(1))


def test_launcher_reports_a_failing_rank(tmp_path):
    from scale_imagenet_amd.launch import spawn_ranks, under_launcher
    assert not under_launcher()
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "if os.environ['RANK'] == '1': sys.exit(7)\n"
                      "time.sleep(30)\n")
    t0 = __import__("time").monotonic()
    assert spawn_ranks([str(script)], 3, timeout_s=60) == 7          # first failure wins; the others are stopped
    assert __import__("time").monotonic() - t0 < 20
    ok = tmp_path / "ok.py"
    ok.write_text("import os\nprint('rank', os.environ['RANK'])\n")
    assert spawn_ranks([str(ok)], 2, timeout_s=60) == 0


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode of the self-launching bench")
def test_bench_self_launch_fails_loudly_without_a_gpu():
    """`python bench.py --gpus 2` without a rank environment starts its own ranks (and must not touch the
    GPU itself to do so); here they find no device, so the parent has to exit non-zero."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "2"], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TTNET_DIST_BACKEND="gloo"))
    assert r.returncode != 0
    assert "HIP device" in r.stderr
