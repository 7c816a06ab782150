import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gisnav_amd.engine import PoseEngine
dev = torch.device("cuda", 0)
eng = PoseEngine(0, max_batch=1, max_kpts=128)
pat = torch.randn(8 * 80 * 256, device=dev)
for blocks, spin in ((64, 1), (256, 1), (2048, 1), (2048, 50), (512, 200)):
    out = torch.zeros(81, dtype=torch.int32, device=dev)
    rc = eng.lib.gn_debug_lds_dma_probe(eng.ctx, C.c_void_p(pat.data_ptr()), C.c_void_p(out.data_ptr()), blocks, spin, eng._stream())
    torch.cuda.synchronize()
    o = out.cpu().tolist()
    bad = {i: v for i, v in enumerate(o[:80]) if v}
    print(f"blocks={blocks} spin={spin} rc={rc} ran={o[80]} mismatching pieces: {bad if bad else 'none'}", flush=True)
