import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from _util import spec_and_state, args_for
from scale_imagenet_amd import ttnet, synth
spec, st = spec_and_state("full")
m = ttnet.TT_vf_19lv3_imgnet(args_for("full"))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
m = m.cuda().eval()
x = torch.from_numpy(synth.synth_images(64)).cuda()
with torch.no_grad():
    y = m(x)
plan = m._any_plan()
pw, dw = plan.query("full_listed_pw"), plan.query("full_listed_dw")
pg = 64 * (2*3136 + 8*841 + 4*841 + 16*256 + 8*256)
do = 64 * 2 * (60*841 + 120*256 + 240*81)
print("listed pixel-groups", pw, "of", pg, pw/pg, " dw outputs", dw, "of", do, dw/do)
