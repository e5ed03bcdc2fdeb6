import time, numpy as np, torch
rt = torch.cuda.cudart()
torch.zeros(1).cuda()
for mb in (368, 2949):
    a = np.ones(mb << 20, np.uint8)
    t0 = time.time(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); t1 = time.time()
    rt.cudaHostUnregister(a.ctypes.data); t2 = time.time()
    print("register %d MB: %.1f ms (%.1f GB/s), unregister %.1f ms, rc=%s" % (mb, (t1-t0)*1e3, a.nbytes/(t1-t0)/1e9, (t2-t1)*1e3, rc))
import sys, os
sys.path.insert(0, os.getcwd())
from helen_amd.engine import HelenEngine
from helen_amd.weights import make_weights
eng = HelenEngine(make_weights(input_scale=1/64.), device=0, max_windows=4096)
img = np.random.default_rng(0).integers(0, 256, (8*4096, 1000, 90), dtype=np.uint8)
eng.polish_host(img[:4096])
t0 = time.time(); eng.polish_host(img); dt = time.time() - t0
print("pageable caller: %.0f windows/s" % (img.shape[0] / dt))
pin = torch.from_numpy(img).pin_memory()
ob = torch.empty((img.shape[0], 1000), dtype=torch.uint8).pin_memory(); orr = torch.empty_like(ob).pin_memory()
eng.polish_host(pin[:4096], out=(ob.numpy()[:4096], orr.numpy()[:4096]))
t0 = time.time(); eng.polish_host(pin, out=(ob.numpy(), orr.numpy())); dt = time.time() - t0
print("page-locked caller: %.0f windows/s" % (img.shape[0] / dt))
