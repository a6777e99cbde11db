"""Run the same forward several times and report the first stage whose bytes differ between runs."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scale_imagenet_amd import synth, ttnet
from tests._util import spec_and_state, args_for

variant = os.environ.get("VARIANT", "small")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cls = {"small": ttnet.TT_vf_19lv3_imgnet_small, "xsmall": ttnet.TT_vf_19lv3_imgnet_xsmall, "full": ttnet.TT_vf_19lv3_imgnet}[variant]
spec, st = spec_and_state(variant)
m = cls(args_for(variant))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
m = m.cuda().eval()
x = torch.from_numpy(synth.synth_images(n)).cuda()
stages = ["features.3", "features.4", "features.5", "flatten"]
ref = None
for it in range(6):
    with torch.no_grad():
        y = m(x).cpu().numpy()
    cur = {s: m.read_stage(s, n).copy() for s in stages}
    cur["logits"] = y
    if ref is None:
        ref = cur
        continue
    for s in stages + ["logits"]:
        a, b = ref[s], cur[s]
        if not np.array_equal(a, b):
            d = np.argwhere(a != b)
            print(f"run {it}: stage {s} differs at {len(d)} positions, first {d[:4].tolist()}", flush=True)
            if s == "logits":
                print("   max |dy| =", np.abs(a - b).max())
            break
    else:
        print(f"run {it}: identical", flush=True)
