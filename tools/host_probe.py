"""What the GPU box's HOST offers to stage D1 and the LAS writer: usable cores, qhull throughput over worker
processes, write rates of one large file under several strategies, D2H into registered shared memory.
python tools/host_probe.py [cores|qhull|write|shm ...]   (default: all)"""
import mmap
import os
import subprocess
import sys
import threading
import time

import numpy as np

WHAT = set(sys.argv[1:]) or {"cores", "qhull", "write", "shm"}


def cores():
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/pids.max"):
        try:
            print(p, open(p).read().strip())
        except OSError as e:
            print(p, "-", e.strerror)
    print(subprocess.run("nproc; df -T /tmp /dev/shm . 2>&1; free -g | head -2; uname -r", shell=True,
                         capture_output=True, text=True).stdout, flush=True)


QH = r"""
import sys, time, numpy as np
from scipy.spatial import ConvexHull
rng = np.random.default_rng(int(sys.argv[1]))
p = (rng.normal(size=(43000, 3)) * [2.5, 2.5, 9]).astype(np.float32).astype(np.float64)
ConvexHull(p, qhull_options="QbB Pp Qt")
sys.stdout.write("r\n"); sys.stdout.flush()
sys.stdin.readline()
t, c = time.perf_counter(), time.process_time()
for _ in range(int(sys.argv[2])):
    ConvexHull(p, qhull_options="QbB Pp Qt")
sys.stdout.write("%f %f\n" % (time.perf_counter() - t, time.process_time() - c)); sys.stdout.flush()
"""


def qhull():
    for procs in (8, 16, 24, 32, 64):
        reps = 24
        ps = [subprocess.Popen([sys.executable, "-c", QH, str(i), str(reps)], stdin=subprocess.PIPE,
                               stdout=subprocess.PIPE, text=True) for i in range(procs)]
        for p in ps:
            p.stdout.readline()
        t = time.perf_counter()
        for p in ps:
            p.stdin.write("go\n")
            p.stdin.flush()
        both = [[float(v) for v in p.stdout.readline().split()] for p in ps]
        each, cpu = [b[0] for b in both], [b[1] for b in both]
        wall = time.perf_counter() - t
        for p in ps:
            p.wait()
        print(f"qhull 43k-point hulls: {procs:4d} processes x {reps}: wall {wall * 1e3:8.1f} ms, "
              f"{procs * reps / wall:8.1f} hulls/s, per hull in a process {1e3 * np.mean(each) / reps:6.2f} ms wall, "
              f"{1e3 * np.mean(cpu) / reps:6.2f} ms CPU", flush=True)


def _pwrite_all(fd, buf, nthreads, block, base=0):
    n = len(buf)
    nb = (n + block - 1) // block
    nxt = [0]
    lock = threading.Lock()
    mv = memoryview(buf)

    def run():
        while True:
            with lock:
                b = nxt[0]
                nxt[0] += 1
            if b >= nb:
                return
            lo = b * block
            hi = min(n, lo + block)
            os.pwrite(fd, mv[lo:hi], base + lo)

    ths = [threading.Thread(target=run) for _ in range(nthreads)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


def write():
    size = 3 << 30
    buf = mmap.mmap(-1, size)                     # page aligned, anonymous
    np.frombuffer(buf, dtype=np.uint8)[::4096] = 1  # touch
    for d in ("/tmp", "/dev/shm"):
        path = os.path.join(d, "pch_probe.bin")
        for label, flags, prealloc, nth, blk in (
                ("buffered 8 thr 8 MB", 0, False, 8, 8 << 20),
                ("buffered fallocate 8 thr 8 MB", 0, True, 8, 8 << 20),
                ("O_DIRECT fallocate 8 thr 8 MB", os.O_DIRECT, True, 8, 8 << 20),
                ("O_DIRECT fallocate 16 thr 32 MB", os.O_DIRECT, True, 16, 32 << 20)):
            try:
                if os.path.exists(path):
                    os.unlink(path)
                t = time.perf_counter()
                fd = os.open(path, os.O_CREAT | os.O_WRONLY | flags, 0o644)
                if prealloc:
                    os.posix_fallocate(fd, 0, size)
                t1 = time.perf_counter()
                _pwrite_all(fd, buf, nth, blk)
                os.close(fd)
                dt = time.perf_counter() - t
                print(f"write 3 GiB {d:9s} {label:34s}: {dt * 1e3:8.1f} ms ({size / dt / 1e9:6.2f} GB/s; "
                      f"open+fallocate {1e3 * (t1 - t):6.1f} ms)", flush=True)
            except OSError as e:
                print(f"write 3 GiB {d:9s} {label:34s}: {e}", flush=True)
        # several files at once (is the limit per inode?)
        try:
            t = time.perf_counter()
            fds = [os.open(path + str(i), os.O_CREAT | os.O_WRONLY, 0o644) for i in range(4)]
            q = size // 4
            ths = [threading.Thread(target=_pwrite_all, args=(fds[i], memoryview(buf)[i * q:(i + 1) * q], 2, 8 << 20))
                   for i in range(4)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            for fd in fds:
                os.close(fd)
            dt = time.perf_counter() - t
            print(f"write 3 GiB {d:9s} {'4 files x 2 thr buffered':34s}: {dt * 1e3:8.1f} ms ({size / dt / 1e9:6.2f} GB/s)",
                  flush=True)
            for i in range(4):
                os.unlink(path + str(i))
        except OSError as e:
            print("4 files:", e)
        if os.path.exists(path):
            os.unlink(path)


def shm():
    import torch
    n = 120 << 20
    dev = torch.device("cuda:0")
    src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(3)
    path = "/dev/shm/pch_probe_shm"
    fd = os.open(path, os.O_CREAT | os.O_RDWR, 0o600)
    os.ftruncate(fd, n)
    m = mmap.mmap(fd, n)
    os.close(fd)
    host = torch.frombuffer(m, dtype=torch.uint8)
    host.fill_(0)
    for label in ("pageable shm", "registered shm"):
        if label.startswith("registered"):
            rc = torch.cuda.cudart().cudaHostRegister(host.data_ptr(), n, 0)
            print("cudaHostRegister rc", rc)
        for r in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            host.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print(f"D2H 120 MiB -> {label}: {dt * 1e3:.2f} ms ({n / dt / 1e9:.1f} GB/s)", flush=True)
    pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    for r in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        pin.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"D2H 120 MiB -> torch pinned: {dt * 1e3:.2f} ms ({n / dt / 1e9:.1f} GB/s)", flush=True)
    torch.cuda.cudart().cudaHostUnregister(host.data_ptr())
    del host
    m.close()
    os.unlink(path)


for name in ("cores", "qhull", "write", "shm"):
    if name in WHAT:
        globals()[name]()
