"""Device time of Resize(256) + CenterCrop(224) (csrc/preproc.hip) and of the whole uint8 pipeline
resize -> crop -> forward_u8 for a batch of decoded images of one size.
usage: python tools/preproc_bench.py [H W] [batch]      (default 375 500, the commonest ImageNet size; 256)"""
import sys, time
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import numpy as np, torch
from _util import spec_and_state, args_for
from scale_imagenet_amd import ttnet, preprocess

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (375, 500)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
dev = torch.device("cuda:0")
spec, st = spec_and_state("small")
m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
m = m.to(dev).eval().reserve(B)
m.set_lanes(2)
rng = np.random.default_rng(0)
raws = [torch.from_numpy(rng.integers(0, 256, size=(B, H, W, 3), dtype=np.uint8)).to(dev) for _ in range(3)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]

def ev_time(fn, reps=50):
    for _ in range(5): fn(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

t_res = ev_time(lambda i: preprocess.resize_center_crop_u8(raws[i % 3]))
with torch.no_grad():
    crop = preprocess.resize_center_crop_u8(raws[0])
    t_fwd = ev_time(lambda i: m.forward_u8(crop))
    def both(i):
        with torch.cuda.stream(streams[i % 2]):
            preprocess.imgnet_eval_forward(m, raws[i % 3], lane=i % 2)
    for i in range(12): both(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 200
    for i in range(K): both(i)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
in_bytes, out_bytes = B * H * W * 3, B * 224 * 224 * 3
print(f"{B} images {H}x{W}: resize + crop {t_res:.1f} us ({(in_bytes + out_bytes) / t_res / 1e3:.0f} GB/s of input + output bytes); "
      f"forward_u8 alone {t_fwd:.1f} us; resize -> crop -> forward_u8, two batches in flight: {K * B / el:,.0f} images/s ({el / K * 1e6:.1f} us per batch)")
