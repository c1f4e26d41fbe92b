"""Do interprocess HIP events work on this stack?  (Round 5: the process form of the row bands drains its stream and meets the other ranks at a
shared-memory barrier at every phase boundary -- a host round trip during which the GPU idles; with hipIpcGetEventHandle / hipIpcOpenEventHandle a
rank's stream could wait for another rank's event as the thread form's streams do.)  Two processes on GPU 0: rank 0 creates an interprocess event,
exports it, records it after a 256 MiB fill; rank 1 opens it, makes its stream wait for it and reads the filled buffer through an IPC mapping.
   python tools/ipc_event_probe.py"""
import ctypes as C
import multiprocessing as mp
import os
import time


def hip():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so")


class Handle(C.Structure):
    _fields_ = [("reserved", C.c_char * 64)]


def worker(rank, conn):
    L = hip()
    L.hipSetDevice(0)
    L.hipIpcGetMemHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    L.hipIpcOpenMemHandle.argtypes = [C.POINTER(C.c_void_p), Handle, C.c_uint]
    L.hipIpcGetEventHandle.argtypes = [C.POINTER(Handle), C.c_void_p]
    L.hipIpcOpenEventHandle.argtypes = [C.POINTER(C.c_void_p), Handle]
    L.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    L.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    L.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    L.hipMemsetAsync.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
    L.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    L.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    L.hipStreamSynchronize.argtypes = [C.c_void_p]
    n = 256 << 20
    s = C.c_void_p(); L.hipStreamCreate(C.byref(s))
    if rank == 0:
        p = C.c_void_p(); rc_m = L.hipMalloc(C.byref(p), C.c_size_t(n))
        ev = C.c_void_p()
        rc_c = L.hipEventCreateWithFlags(C.byref(ev), 0x2 | 0x4)          # hipEventDisableTiming | hipEventInterprocess
        hm, he = Handle(), Handle()
        rc_gm = L.hipIpcGetMemHandle(C.byref(hm), p)
        rc_ge = L.hipIpcGetEventHandle(C.byref(he), ev)
        print(f"rank 0: malloc {rc_m}, event create (interprocess) {rc_c}, mem handle {rc_gm}, hipIpcGetEventHandle rc {rc_ge}", flush=True)
        conn.send((bytes(hm), bytes(he), rc_ge))
        conn.recv()                                   # rank 1 has opened both
        for rep in range(3):
            L.hipMemsetAsync(p, 0x11 * (rep + 1), n, s)
            rc_r = L.hipEventRecord(ev, s)
            conn.send(("recorded", rc_r))
            conn.recv()
        L.hipStreamSynchronize(s)
    else:
        hm, he, rc_ge = conn.recv()
        q = C.c_void_p(); ev = C.c_void_p()
        rc_om = L.hipIpcOpenMemHandle(C.byref(q), Handle.from_buffer_copy(hm), 1)
        rc_oe = L.hipIpcOpenEventHandle(C.byref(ev), Handle.from_buffer_copy(he)) if rc_ge == 0 else -1
        print(f"rank 1: hipIpcOpenMemHandle rc {rc_om}, hipIpcOpenEventHandle rc {rc_oe}", flush=True)
        conn.send("opened")
        host = C.create_string_buffer(16)
        for rep in range(3):
            _, rc_r = conn.recv()
            t0 = time.perf_counter()
            rc_w = L.hipStreamWaitEvent(s, ev, 0) if rc_oe == 0 else -1
            rc_cp = L.hipMemcpyAsync(host, C.c_void_p(q.value + n - 16), 16, 2, s) if rc_om == 0 else -1
            L.hipStreamSynchronize(s)
            dt = (time.perf_counter() - t0) * 1e6
            print(f"rank 1 rep {rep}: record rc {rc_r}, hipStreamWaitEvent rc {rc_w}, last bytes of the buffer after the wait: {host.raw[:4].hex()} "
                  f"(expected {'%02x' % (0x11 * (rep + 1)) * 4}), {dt:.0f} us", flush=True)
            conn.send("ok")


def main():
    ctx = mp.get_context("spawn")
    a, b = ctx.Pipe()
    ps = [ctx.Process(target=worker, args=(0, a)), ctx.Process(target=worker, args=(1, b))]
    for p in ps:
        p.start()
    for p in ps:
        p.join(120)


if __name__ == "__main__":
    main()
