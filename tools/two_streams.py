"""Experiment: R model replicas on R HIP streams, each forwarding batches of B/parts images
(graph replay, so the host is not the limit): does overlap of kernel ramps / tails across
streams raise images/s over one stream?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from argparse import Namespace
import torch
from scale_imagenet_amd import synth, ttnet
from scale_imagenet_amd.spec import make_spec
dev = torch.device("cuda", 0)
spec = make_spec("small"); st = synth.synth_state_dict(spec)
def mk(b):
    m = ttnet.TT_vf_19lv3_imgnet_small(Namespace(nfilter=8, tfilter=8, layers=1, groups=[1, None, 4, None]))
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
    return m.to(dev).eval().reserve(b)
B = 256
x = torch.from_numpy(synth.synth_images(B)).to(dev)
for replicas, per in ((1, 256), (2, 256), (3, 256), (2, 128), (4, 64)):
    ms = [mk(per) for _ in range(replicas)]
    streams = [torch.cuda.Stream(dev) for _ in range(replicas)]
    xi = x[:per]
    def step():
        for m, s in zip(ms, streams):
            with torch.cuda.stream(s), torch.no_grad():
                m(xi)
    for _ in range(6): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 40
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{replicas} stream(s) x batch {per}: {dt * 1e3:.4f} ms per {replicas * per} images -> {replicas * per / dt:.0f} img/s", flush=True)
    del ms
