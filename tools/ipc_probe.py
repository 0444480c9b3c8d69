"""Bring-up probe: can two processes on this box map each other's device memory with hipIpc (plain and uncached allocations),
store into it from a kernel-less path (hipMemcpy) and see the data?  python tools/ipc_probe.py"""
import ctypes as C, multiprocessing as mp, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def child(q_in, q_out):
    from sobfu_amd import _lib
    L = _lib.lib()
    hip = C.CDLL("libamdhip64.so")
    for name in ("plain", "uncached"):
        h = q_in.get()
        p = C.c_void_p()
        rc = L.sobfu_hip_ipc_open((C.c_char * 64).from_buffer_copy(h), C.byref(p))
        if rc == 0:
            v = (C.c_uint32 * 4)(11, 22, 33, 44)
            hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
            rc2 = hip.hipMemcpy(p, v, 16, 1)
            hip.hipDeviceSynchronize()
            q_out.put((name, rc, rc2))
            q_in.get()
            L.sobfu_hip_ipc_close(p)
        else:
            q_out.put((name, rc, None))


if __name__ == "__main__":
    mp.set_start_method("spawn")
    from sobfu_amd import _lib
    L = _lib.lib()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    qi, qo = mp.Queue(), mp.Queue()
    pr = mp.Process(target=child, args=(qi, qo)); pr.start()
    for name, flag in (("plain", None), ("uncached", 0x3)):
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), 4096) if flag is None else hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(4096), C.c_uint(flag))
        h = (C.c_char * 64)()
        rce = L.sobfu_hip_ipc_export(p, h)
        print(f"{name}: alloc rc={rc} export rc={rce}", flush=True)
        qi.put(bytes(h))
        got = qo.get(timeout=60)
        v = (C.c_uint32 * 4)()
        hip.hipMemcpy(v, p, 16, 2)
        print(f"{name}: child open/copy -> {got}; parent reads {list(v)}", flush=True)
        if got[1] == 0:
            qi.put(b"")
    pr.join(timeout=30)
