"""Save / load rate of the prepared-gallery file (MI355GAL v2: checksummed sections through two pinned buffers).
Usage on the GPU box: python scripts/gallery_io_rate.py [rows] -> one JSON line (kept under profiles/)."""
import json, os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import isehr_amd  # noqa: F401
from isehr_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1005994
d = 2048
raw = torch.empty((n, d), dtype=torch.float32, device="cuda")
_lib.synth_fill_device(raw.data_ptr(), 1234, 0, n, d, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
g = _lib.Gallery.from_device_ptr(raw.data_ptr(), n, d)
del raw
path = os.path.join(tempfile.gettempdir(), "mi355_gallery_rate.bin")
t0 = time.time(); g.save(path); t_save = time.time() - t0
size = os.path.getsize(path)
g.close()
t0 = time.time(); h = _lib.Gallery.load(path); t_load = time.time() - t0      # page cache warm (just written)
h.close()
t0 = time.time(); h = _lib.Gallery.load(path); t_load = min(t_load, time.time() - t0)
h.close()
os.remove(path)
print(json.dumps({"rows": n, "dim": d, "file_bytes": size, "save_s": round(t_save, 3), "save_GBps": round(size / t_save / 1e9, 2),
                  "load_s": round(t_load, 3), "load_GBps": round(size / t_load / 1e9, 2),
                  "note": "load = 4 reader threads pread() into a ring of 8 pinned 32 MiB buffers, hipMemcpyAsync per chunk + device-side section "
                          "checksums; file in the page cache"}))
