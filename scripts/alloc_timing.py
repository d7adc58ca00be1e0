import ctypes, time, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
import isehr_amd
from isehr_amd import _lib
hip = ctypes.CDLL('libamdhip64.so')
torch.cuda.init(); torch.cuda.synchronize()
def t_malloc(nbytes):
    p = ctypes.c_void_p()
    t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)); t1 = time.perf_counter()
    hip.hipDeviceSynchronize()
    t2 = time.perf_counter(); hip.hipFree(p); t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t3 - t2) * 1e3
for rep in range(3):
    for nb in (8241102848, 4120551424, 12072960, 64 << 20):
        a, f = t_malloc(nb)
        print('hipMalloc %6.0f MB: %.3f ms, hipFree %.3f ms' % (nb / 1e6, a, f))
n, d = 1005994, 2048
raw = torch.empty((n, d), dtype=torch.float32, device='cuda')
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d, norm_mode=_lib.NORM_L2, device=0)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    g.close()
    t2 = time.perf_counter()
    print('create %.3f ms, close %.3f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
