import time, numpy as np, torch, mmap, os, tempfile
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
rt = torch.cuda.cudart()
n = 4096 * 116000
for trial in range(2):
    fd, path = tempfile.mkstemp(dir="/dev/shm"); os.ftruncate(fd, n)
    m = mmap.mmap(fd, n); a = np.frombuffer(m, dtype=np.uint8)
    t0 = time.time(); a[::4096] = 1; t1 = time.time()
    rc = rt.cudaHostRegister(a.ctypes.data, n, 0); t2 = time.time()
    rt.cudaHostUnregister(a.ctypes.data); t3 = time.time()
    del a; m.close(); os.close(fd); os.unlink(path); t4 = time.time()
    print("touch %.3f  register %.3f (rc %s)  unregister %.3f  unmap+unlink %.3f  for %.0f MB" % (t1-t0, t2-t1, rc, t3-t2, t4-t3, n/1e6))
