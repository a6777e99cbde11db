"""CPU: what ``nn.DataParallel`` does to the module over several devices (main.py:192 wraps over all visible
GPUs), without GPUs: ``replicate`` shallow-copies the module per device and forward.  The replicas must share
the original's per-device plan cache under ONE lock and track the ORIGINAL's weights (their own parameters are
fresh broadcast copies on every forward), so that a weight update reloads every device's plan and an unchanged
model never rebuilds its truth tables."""
import copy
import threading

import torch

from _util import args_for
from scale_imagenet_amd import ttnet


def _model():
    return ttnet.TT_vf_19lv3_imgnet_small(args_for("small")).eval()


def test_replicas_share_the_plan_cache_and_follow_the_original():
    m = _model()
    r1, r2 = m._replicate_for_data_parallel(), m._replicate_for_data_parallel()
    assert r1.__dict__["_sig_source"] is m and r2.__dict__["_sig_source"] is m
    assert r1._plans is m._plans and r2._plans is m._plans                      # one plan per device, kept across forwards
    assert r1.__dict__["_plans_lock"] is m.__dict__["_plans_lock"] is r2.__dict__["_plans_lock"]
    # replicate() then gives the replica fresh parameter copies: its signature must not come from those
    for mod in r1.modules():
        for k in list(mod._parameters):
            if mod._parameters[k] is not None:
                mod._parameters[k] = mod._parameters[k].detach().clone()
    sig0 = m._state_signature()
    assert r1._state_signature() == sig0 == r2._state_signature()
    with torch.no_grad():
        m.features[1].weight.mul_(1.0)                                           # in-place update of the original
    sig1 = m._state_signature()
    assert sig1 != sig0 and r1._state_signature() == sig1 == r2._state_signature()
    # a replica of a replica still points at the original
    assert r1._replicate_for_data_parallel().__dict__["_sig_source"] is m


def test_plan_lookup_is_serialised():
    """_plan_for takes the shared lock (parallel_apply runs replicas on threads): with the lock held elsewhere
    a second caller blocks instead of racing on the cache."""
    m = _model()
    lock = m.__dict__["_plans_lock"]
    entered = threading.Event()

    def worker():
        try:
            m._plan_for(torch.device("cuda", 0), 1)                              # no HIP device here: fails AFTER taking the lock
        except Exception:
            pass
        entered.set()

    with lock:
        t = threading.Thread(target=worker, daemon=True)
        t.start()
        assert not entered.wait(0.3)                                             # blocked on the lock
    assert entered.wait(30)
    t.join(30)


def test_deepcopy_leaves_plans_behind():
    m = _model()
    m._plans[0] = object()                                                       # stands for a device plan
    c = copy.deepcopy(m)
    assert c._plans == {} and c.__dict__["_plans_lock"] is not m.__dict__["_plans_lock"]
    assert c.__dict__.get("_sig_source") is None
    assert list(c.state_dict().keys()) == list(m.state_dict().keys())
    m._plans.clear()
