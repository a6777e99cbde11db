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
n = 1
rng = np.random.default_rng(7)
bits = rng.integers(0, 2, size=(n, 64, 56, 56), dtype=np.uint8)
rows_t = torch.from_numpy(OB.pack_rows(bits).view(np.int64)).to(dev)
with torch.no_grad():
    m.forward_from_stem_bits(rows_t)
blk = spec.blocks[0]
tab = m.get_table(blk.conv1.name)          # [64, 65536, 1]
idx = OB.window_index(bits, blk.conv1)     # [1, 64, 29, 29]
got = OB.unpack_rows(m.read_stage("features.4.out1", n), 29)[0]   # [64,29,29]
exp = np.stack([tab[c, idx[0, c], 0] for c in range(64)])
print("mismatch", (got != exp).sum())
# hypothesis: lane reads table of channel ct with rows of channel cr
for c in (1, 2, 17):
    for oy in (0, 1, 2):
        best = []
        for ct in range(64):
            for cr in range(64):
                e = tab[ct, idx[0, cr, oy], 0]
                if np.array_equal(e, got[c, oy]):
                    best.append((ct, cr))
        print("channel", c, "row", oy, "matches (table, rows):", best[:6])
# hypothesis: index shifted
c, oy = 1, 0
print("got", got[c, oy]); print("exp", exp[c, oy])
for sh in range(-3, 4):
    e = tab[c, np.clip(idx[0, c, oy] , 0, 65535), 0]
