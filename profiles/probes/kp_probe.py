import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import synth
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), f"libkp_probe_{sys.argv[1]}.so"))
from macvo_amd import _lib as L
H, W = 480, 640
dev = "cuda"
cam, frames, _ = synth.make_sequence(2, H, W, C=16, iters=1, seed=1000, closed_loop=True)
from macvo_amd import ops
maps = ops.frontend_epilogue(frames[1]["flow"].to(dev), frames[1]["logcov"].to(dev), cam["baseline"], cam["fx"])
fc = maps.flow_cov
p = L.mvKpSelectParams(H, W, 0, 7, 32, 0.0, 0.0, 100.0)
lib.mv_kp_select_workspace_bytes.restype = C.c_size_t
nb = lib.mv_kp_select_workspace_bytes(H, W)
ws = torch.zeros(nb // 8 + 1, dtype=torch.int64, device=dev)
cand = torch.empty(H * W, dtype=torch.int32, device=dev); cnt = torch.empty(4, dtype=torch.int32, device=dev); st = torch.empty(4, device=dev)
lib.mv_kp_select.argtypes = [C.c_void_p] * 7 + [C.POINTER(L.mvKpSelectParams), C.c_void_p, C.c_size_t] + [C.c_void_p] * 4
for _ in range(20):
    lib.mv_kp_select(fc.data_ptr(), None, None, None, None, None, None, C.byref(p), ws.data_ptr(), nb, cand.data_ptr(), cnt.data_ptr(), st.data_ptr(), None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.mv_kp_select(fc.data_ptr(), None, None, None, None, None, None, C.byref(p), ws.data_ptr(), nb, cand.data_ptr(), cnt.data_ptr(), st.data_ptr(), None)
e1.record(); torch.cuda.synchronize()
print("nms + finish back-to-back: %.2f us per select" % (e0.elapsed_time(e1) * 1e3 / 50))
for _ in range(2):
    rc = lib.mv_kp_select(fc.data_ptr(), None, None, None, None, None, None, C.byref(p), ws.data_ptr(), nb, cand.data_ptr(), cnt.data_ptr(), st.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    buf = (C.c_longlong * 16)()
    lib.kp_probe_stamps(buf)
    t = list(buf)[:6]
    print("count", cnt.tolist(), "phases (us): load", (t[1]-t[0])/100, "median", (t[2]-t[1])/100, "thresh", (t[3]-t[2])/100, "count+scan", (t[4]-t[3])/100, "emit", (t[5]-t[4])/100, "total", (t[5]-t[0])/100)
