"""Cost of device allocations of matrix size on the box (hipMalloc / hipFree / first touch), via torch's HIP runtime."""
import ctypes, time, sys
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
hip.hipFree.argtypes = [ctypes.c_void_p]
hip.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
hip.hipDeviceSynchronize()
for mb in (16, 64, 256, 640, 1024, 2048):
    for rep in range(2):
        p = ctypes.c_void_p()
        t0 = time.perf_counter(); rc = hip.hipMalloc(ctypes.byref(p), mb << 20); t1 = time.perf_counter()
        hip.hipMemset(p, 0, mb << 20); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        print("%5d MB  hipMalloc %.2f ms  first memset %.2f ms  hipFree %.2f ms  (rc %d)" % (mb, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, rc))
