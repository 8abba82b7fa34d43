import os, sys, ctypes as C, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import bench, oracle_lib as ol
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/kernel/mm/transparent_hugepage/enabled"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "-", e)
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|NUMA|MHz' | head -12")
L = ol.lib(); H, W, Cc = bench.H, bench.W, bench.C
x = ol.synth_f32(bench.SEED, H * W * Cc)
cpus, per = bench.physical_core_cpus()
fn = L.cpu_port_roundtrip_bench_f32; fn.restype = C.c_int
fn.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
for n in (1, 2, 4, 8, 16, 32, 64, 128):
    if n > len(cpus): break
    for pinned in (True, False):
        pin = (C.c_int * n)(*cpus[:n]) if pinned else None
        sec, err = C.c_double(), C.c_double()
        reps = max(1, n // 2)
        rc = fn(H, W, Cc, x.ctypes.data, n, pin, reps, C.byref(sec), C.byref(err))
        print(f"threads {n:4d} pinned {pinned!s:5}: rc {rc} {reps * H * W / 1e6 / sec.value:8.1f} Mpix/s  ({sec.value / reps * 1e3:7.1f} ms per roundtrip)", flush=True)
