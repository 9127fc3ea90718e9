import sys, ctypes; sys.path.insert(0, '/root/repo')
import numpy as np, mixedprecisionblockqr_amd as mp
hip = ctypes.CDLL("libamdhip64.so")
def chk(tag):
    n = ctypes.c_int(-1); rc = hip.hipGetDeviceCount(ctypes.byref(n)); le = hip.hipGetLastError()
    free = ctypes.c_size_t(); tot = ctypes.c_size_t(); rc2 = hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(tot))
    print(tag, "hipGetDeviceCount rc", rc, "n", n.value, "lastError", le, "memGetInfo rc", rc2, "free GB", free.value / 1e9, flush=True)
chk("start")
hh = mp.Handle(0)
m, n, r = 65536, 8192, 256
prec = mp.PREC_FP8 if (len(sys.argv) < 2 or sys.argv[1] == "fp8") else mp.PREC_FP16
hh.plan(m, n, r, precision=prec); chk("planned")
hh.generate(1234); hh.factor(); hh.sync(); chk("factored")
if len(sys.argv) > 2: print(hh.metrics()); chk("metrics")
hh.close(); chk("closed")
import torch
try:
    torch.cuda.init(); print("torch ok", torch.cuda.device_count())
except Exception as e:
    print("torch failed:", e)
chk("end")
