import sys, time, torch, numpy as np, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gisnav_amd import loftr_synthetic as ls
from gisnav_amd.loftr import LoFTR
sd = ls.synthetic_state_dict(0); i0, i1 = ls.synthetic_pair(1, 480, 640)
d = {"image0": i0.cuda(), "image1": i1.cuda()}
res = {}
for var in (103, 106):
    m = LoFTR(state_dict=sd, graph=False).to("cuda:0").eval()
    out = m(d, with_ids=True)
    m.lib.gn_loftr_set_graph(m._ctx, var)
    for _ in range(3): out = m(d, with_ids=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): out = m(d, with_ids=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    tok = m.debug_read("tok", 2 * 4864 * 256).copy()
    res[var] = (out, tok)
    print(var, "ms", dt * 1e3, "matches", out["keypoints0"].shape[0])
a, b = res[103], res[106]
print("same ids", torch.equal(a[0]["i_ids"], b[0]["i_ids"]) and torch.equal(a[0]["j_ids"], b[0]["j_ids"]), "tok rel", float(np.abs(a[1] - b[1]).max() / np.abs(a[1]).max()),
      "conf", float((a[0]["confidence"] - b[0]["confidence"]).abs().max()), "k1", float((a[0]["keypoints1"] - b[0]["keypoints1"]).abs().max()))
