import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
envi = importlib.import_module("doudizhu-rl_amd.envi")
e = envi.EnvCooperationSimplify(seed=0)
e.reset(); e.prepare()
def tm(name, f, n=2000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter()-t0)/n*1e6:7.2f} us")
C = e._C
tm("torch.empty (6,15,4)", lambda: torch.empty((6,15,4), dtype=torch.float32, device=e.device))
tm("torch.empty 1-D", lambda: torch.empty((3000,), dtype=torch.float32, device=e.device))
buf = torch.empty((40*60,), dtype=torch.float32, device=e.device)
tm("slice+view", lambda: buf[:360].view(6,15,4))
tm("raw stream", lambda: e._raw(e._di))
tm("c_void_p", lambda: C.c_void_p(12345))
tm("ddz_sync only", lambda: e._L.ddz_sync(e._di, C.c_void_p(e._raw(e._di))))
out = torch.empty((6,15,4), dtype=torch.float32, device=e.device)
tm("ddz_observe call (async)", lambda: e._L.ddz_observe(e._h, 3, C.c_void_p(out.data_ptr()), C.c_void_p(e._raw(e._di))), 500)
def obs_sync():
    e._L.ddz_observe(e._h, 3, C.c_void_p(out.data_ptr()), C.c_void_p(e._raw(e._di)))
    e._L.ddz_sync(e._di, C.c_void_p(e._raw(e._di)))
tm("ddz_observe + sync", obs_sync)
def face(): return e.face
tm("e.face", face, 500)
tm("e.valid_actions()", lambda: e.valid_actions(), 500)
import random
tm("random.randrange", lambda: random.randrange(30))
tm("get_role_ID+get_curr_handcards", lambda: (e.get_role_ID(), e.get_curr_handcards()))
def ply():
    _, done, _ = e.step_random()
    if done:
        e.reset(); e.prepare()
tm("step_random (+reset)", ply, 1000)
def full():
    f = e.face; a = e.valid_actions(); ply()
tm("full ply", full, 1000)
