"""Debug helper (GPU box): compare every stage of the HIP path with the bit oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from _util import args_for, spec_and_state
from oracle import ttnet_bits as OB
from scale_imagenet_amd import synth, ttnet

dev = torch.device("cuda", 0)
spec, st = spec_and_state("small")
m = ttnet.TT_vf_19lv3_imgnet_small(args_for("small"))
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()})
m = m.to(dev).eval().reserve(8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(7)
bits = rng.integers(0, 2, size=(n, 64, 56, 56), dtype=np.uint8)
rows_t = torch.from_numpy(OB.pack_rows(bits).view(np.int64)).to(dev)
with torch.no_grad():
    y = m.forward_from_stem_bits(rows_t).cpu().numpy()
luts = {b.name: m.get_table(b.name) for b in spec.block_tts()}
bt = {}
ref = OB.forward_from_stem_bits(bits, st, spec, luts, bt)
for stage in bt:
    if stage in ("flatten", "features.6"):
        continue
    got = m.read_stage(stage, n)
    w = bt[stage].shape[-1]
    gb = OB.unpack_rows(got, w)
    d = np.argwhere(gb != bt[stage])
    print(stage, bt[stage].shape, "mismatches", len(d), "first", d[:5].tolist())
    if len(d) and stage.endswith("4.out1"):
        print("   by image", np.bincount(d[:, 0], minlength=n), "by row", np.bincount(d[:, 2], minlength=w)[:32], "by col", np.bincount(d[:, 3], minlength=w)[:32])
        print("   by channel%16", np.bincount(d[:, 1] % 16, minlength=16))
print("logits diff", np.abs(y - ref).max())
