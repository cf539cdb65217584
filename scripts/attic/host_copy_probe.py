"""D2H / H2D of a top-k result / a feed: pageable vs torch-pinned vs hipHostRegister'ed numpy memory (copy + host touch)."""
import time
import numpy as np
import torch
dev = torch.device("cuda", 0)
idx = torch.randint(0, 140000, (1024, 500), dtype=torch.int32, device=dev)
torch.cuda.synchronize()


def t(fn, n=50):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("D2H 2 MB pageable .cpu().numpy(): %.0f us" % t(lambda: idx.cpu().numpy()))
pin = torch.empty((1024, 500), dtype=torch.int32, pin_memory=True)
def d2h_pin():
    pin.copy_(idx, non_blocking=True); torch.cuda.current_stream().synchronize()
print("D2H 2 MB torch-pinned copy_+sync: %.0f us" % t(d2h_pin))
print("   host read of the pinned buffer (sum): %.0f us" % t(lambda: int(pin.numpy().sum()), 10))
print("   host copy out of the pinned buffer (np.copy): %.0f us" % t(lambda: pin.numpy().copy(), 10))
reg = np.empty((1024, 500), np.int32)
rt = torch.from_numpy(reg)
rc = torch.cuda.cudart().cudaHostRegister(reg.ctypes.data, reg.nbytes, 0)
print("cudaHostRegister rc:", rc, "is_pinned:", rt.is_pinned())
def d2h_reg():
    rt.copy_(idx, non_blocking=True); torch.cuda.current_stream().synchronize()
print("D2H 2 MB host-registered numpy copy_+sync: %.0f us" % t(d2h_reg))
print("   host read of it (sum): %.0f us" % t(lambda: int(reg.sum()), 10))
print("   host copy out of it: %.0f us" % t(lambda: reg.copy(), 10))
src = np.random.randint(0, 1000, (60000, 2)).astype(np.int64)       # ~1 MB feed
print("H2D 1 MB pageable: %.0f us" % t(lambda: torch.from_numpy(src).to(dev)))
regs = np.empty_like(src); rts = torch.from_numpy(regs)
torch.cuda.cudart().cudaHostRegister(regs.ctypes.data, regs.nbytes, 0)
dst = torch.empty((60000, 2), dtype=torch.int64, device=dev)
def h2d_reg():
    np.copyto(regs, src); dst.copy_(rts, non_blocking=True)
print("H2D 1 MB via host-registered staging (np.copyto + async copy): %.0f us host time" % t(h2d_reg))
print("   np.copyto alone: %.0f us" % t(lambda: np.copyto(regs, src), 20))
pins = torch.empty((60000, 2), dtype=torch.int64, pin_memory=True)
print("   np.copyto into torch-pinned: %.0f us" % t(lambda: np.copyto(pins.numpy(), src), 10))
