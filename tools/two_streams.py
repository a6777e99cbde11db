"""Experiment: does running two half-batches concurrently on two HIP streams beat one full batch?"""
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
for parts in (1, 2, 4):
    ms = [mk(B // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream(dev) for _ in range(parts)]
    xs = list(x.chunk(parts))
    def step():
        for m, s, xi in zip(ms, streams, xs):
            with torch.cuda.stream(s), torch.no_grad():
                m(xi)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); K = 40
    for _ in range(K): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f"{parts} stream(s) x batch {B // parts}: {dt * 1e3:.4f} ms per {B} images -> {B / dt:.0f} img/s")
