"""How fast can a pageable numpy array get into HBM (and labels back)?  pageable copy, register-in-place, threaded pinned staging."""
import time, sys, threading
import numpy as np, torch
n = 10061824
a = np.random.default_rng(0).random((n, 3))
def t(f, reps=3):
    f(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
dst = torch.empty((n, 3), dtype=torch.float64, device="cuda")
print("pageable .copy_ : %.1f ms" % t(lambda: dst.copy_(torch.from_numpy(a))))
rt = torch.cuda.cudart()
def reg():
    rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
    dst.copy_(torch.from_numpy(a), non_blocking=True); torch.cuda.synchronize()
    rt.cudaHostUnregister(a.ctypes.data)
print("register + copy + unregister : %.1f ms" % t(reg))
CH = 16 << 20
def threaded(nthreads):
    flat = a.reshape(-1).view(np.uint8); dflat = dst.view(torch.uint8).reshape(-1)
    nb = flat.shape[0]; chunks = [(o, min(CH, nb - o)) for o in range(0, nb, CH)]
    pins = [[torch.empty(CH, dtype=torch.uint8).pin_memory() for _ in range(2)] for _ in range(nthreads)]
    streams = [torch.cuda.Stream() for _ in range(nthreads)]
    def work(k):
        ev = [None, None]
        with torch.cuda.stream(streams[k]):
            for j, (o, ln) in enumerate(chunks[k::nthreads]):
                b = j & 1
                if ev[b] is not None: ev[b].synchronize()
                np.copyto(pins[k][b].numpy()[:ln], flat[o:o + ln])
                dflat[o:o + ln].copy_(pins[k][b][:ln], non_blocking=True)
                e = torch.cuda.Event(); e.record(streams[k]); ev[b] = e
    def run():
        th = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        [x.start() for x in th]; [x.join() for x in th]
    return run
for nt in (1, 2, 4, 8):
    print("pinned staging, %d threads : %.1f ms" % (nt, t(threaded(nt))))
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
print("labels pageable .cpu() : %.1f ms" % t(lambda: lab.cpu()))
pin = torch.empty(n, dtype=torch.int32).pin_memory()
print("labels into pinned : %.1f ms" % t(lambda: pin.copy_(lab, non_blocking=True)))
