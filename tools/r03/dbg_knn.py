import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hipporag_amd import _lib
from hipporag_amd._lib import check
lib = _lib.load(); dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(7)
for dim in (96, 128):
    keys = rng.standard_normal((3200, dim)).astype(np.float32); q = keys[:50].copy()
    kt = torch.from_numpy(keys).to(dev); qt = torch.from_numpy(q).to(dev)
    ks = torch.empty((3200, 3*dim), dtype=torch.int16, device=dev); qs = torch.empty((50, 3*dim), dtype=torch.int16, device=dev)
    check(lib.hrag_split_f32(kt.data_ptr(), 3200, dim, 0, 1, ks.data_ptr(), st)); check(lib.hrag_split_f32(qt.data_ptr(), 50, dim, 1, 1, qs.data_ptr(), st))
    wsb = int(lib.hrag_sim_topk_workspace_bytes(3200, 50)); ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)
    i16 = torch.empty((50, 16), dtype=torch.int32, device=dev); v16 = torch.empty((50, 16), dtype=torch.float32, device=dev)
    check(lib.hrag_sim_topk(ks.data_ptr(), 3200, 3*dim, qs.data_ptr(), 50, 16, 1, ws.data_ptr(), wsb, i16.data_ptr(), v16.data_ptr(), st))
    sc = torch.empty((50, 3200), dtype=torch.float32, device=dev)
    check(lib.hrag_sim_gemm(ks.data_ptr(), 3200, 3*dim, qs.data_ptr(), 50, sc.data_ptr(), 3200, 0, 1, st))
    torch.cuda.synchronize()
    top = torch.topk(sc, 16, dim=1)
    print(dim, "fused", v16[0, :4].tolist(), i16[0, :4].tolist(), "dense", top.values[0, :4].tolist(), top.indices[0, :4].tolist(), "wsb", wsb)
