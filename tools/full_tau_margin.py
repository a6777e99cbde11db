"""How much margin the full variant's error bound has: bits of the fast path with the bound scaled down
(TTNET_FULL_TAU_SCALE) against the all-float64 path (TTNET_FULL_EXACT=1), on N synthetic images."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from _util import spec_and_state, args_for
from scale_imagenet_amd import ttnet, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spec, st = spec_and_state("full")
m = ttnet.TT_vf_19lv3_imgnet(args_for("full"))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
m = m.cuda().eval().reserve(N)
x = torch.from_numpy(synth.synth_images(N)).cuda()
stages = [b.name for b in spec.blocks[:-1]]
def run():
    with torch.no_grad():
        m(x)
    return {s: m.read_stage(s, N).copy() for s in stages}
os.environ["TTNET_NO_GRAPH"] = "1"
os.environ["TTNET_FULL_EXACT"] = "1"
exact = run()
os.environ["TTNET_FULL_EXACT"] = "0"
plan = m._any_plan()
for scale in ("1", "0.5", "0.25", "0.1", "0.03", "0.01", "0.003"):
    os.environ["TTNET_FULL_TAU_SCALE"] = scale
    p0, d0 = plan.query("full_listed_pw"), plan.query("full_listed_dw")
    got = run()
    diff = {s: int(np.unpackbits((got[s] ^ exact[s]).view(np.uint8)).sum()) for s in stages}
    print(f"tau x {scale}: listed {plan.query('full_listed_pw') - p0} pixel-groups, {plan.query('full_listed_dw') - d0} depthwise outputs; differing bits per stage {diff}")
