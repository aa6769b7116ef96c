"""Does the library's streaming-read probe depend on WHAT it reads?  (round 6: the same kernel geometry reads 6.4 TB/s in
tools/ubench/read_shape.hip on a low-entropy fill and 5.2 TB/s in bench.py on normal_() data)
usage (GPU box): python tools/gpu_probe_data_dependence.py"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from dqc_amd import lib

dev = torch.device("cuda:0")
n = (2 << 30) // 8


def rate(buf, reps=10):
    lib.probe_stream_read(buf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.probe_stream_read(buf)
    e1.record()
    torch.cuda.synchronize()
    return reps * buf.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9


buf = torch.empty(n, dtype=torch.float64, device=dev)
for label, fill in [("zeros", lambda: buf.zero_()),
                    ("ubench pattern 1e-3 (i % 977)", lambda: buf.copy_((torch.arange(n, device=dev) % 977).double() * 1e-3)),
                    ("normal_()", lambda: buf.normal_()),
                    ("uniform random bits", lambda: buf.view(torch.int64).random_()),
                    ("zeros again", lambda: buf.zero_())]:
    fill()
    torch.cuda.synchronize()
    print("%-34s %6.0f %6.0f %6.0f GB/s" % (label, rate(buf), rate(buf), rate(buf)))
