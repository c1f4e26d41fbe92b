"""What a host-staged call of octane_amd/exchange.py costs on the GPU box, piece by piece (round 4: the collective-transport tests took ~85 ms per
exchange call where gloo itself needs 0.5-2 ms): aliasing a device pointer as a tensor, the device -> host copy, host -> device, the synchronisation."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from octane_amd import exchange

def t(f, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

dev = torch.device("cuda", 0)
base = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
ptr, nb = base.data_ptr(), 114688
mk = lambda: torch.as_tensor(exchange._DevMem(ptr, nb), device=dev)
print(f"as_tensor(__cuda_array_interface__)      {t(mk):8.3f} ms")
x = mk()
print(f"aliased.cpu() of {nb} B                 {t(lambda: x.cpu()):8.3f} ms")
h = x.cpu()
print(f"aliased.copy_(host)                      {t(lambda: x.copy_(h)):8.3f} ms")
print(f"native slice .cpu()                      {t(lambda: base[:nb].cpu()):8.3f} ms")
print(f"torch.cuda.synchronize                   {t(lambda: torch.cuda.synchronize(dev)):8.3f} ms")
import ctypes as C
from octane_amd import capi
capi.lib()
hip = C.CDLL(None)
buf = (C.c_ubyte * nb)()
try:
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    print(f"hipMemcpy D2H of {nb} B through ctypes   {t(lambda: hip.hipMemcpy(buf, ptr, nb, 2)):8.3f} ms")
    print(f"hipMemcpy H2D                            {t(lambda: hip.hipMemcpy(ptr, buf, nb, 1)):8.3f} ms")
except Exception as e:
    print("hipMemcpy through ctypes not available:", e)
