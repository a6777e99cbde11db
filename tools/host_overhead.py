"""Host enqueue time per forward vs device time per forward (is the step launch-bound?)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scale_imagenet_amd import synth, ttnet
from tests._util import spec_and_state, args_for

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec, st = spec_and_state("small")
m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(synth.synth_images(n)).cuda()
with torch.no_grad():
    for _ in range(5):
        m(x)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            m(x)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"B={n}: host enqueue {1e6 * (t1 - t0) / 50:.1f} us/forward, total {1e6 * (t2 - t0) / 50:.1f} us/forward", flush=True)
