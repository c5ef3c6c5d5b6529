"""What ONE small collective costs on the critical path of a kernel chain (GPU only; a world of one on RCCL: the collective is the
identity, what is measured is c10d's launch path -- stream hops, events, host time):
    python tools/micro/collective_chain.py
chain = [tiny kernel -> (all_reduce of 2 KB) -> tiny kernel] x N, timed with HIP events and host clocks."""
import os, time, sys
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
N = 300
a = torch.zeros(256, dtype=torch.float64, device=dev)
b = torch.zeros(1 << 14, device=dev)


def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(60_000_000)          # (the host runs ahead: the chain below is timed GPU-side, not at the host's pace)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        b.add_(1.0)
        if mode == "sync":
            dist.all_reduce(a)
        elif mode == "async":
            w = dist.all_reduce(a, async_op=True)
            w.wait()
        elif mode == "hop":            # what a cross-stream round trip alone costs
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                a.add_(0.0)
            torch.cuda.current_stream().wait_stream(s)
        a.add_(1.0)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3, (t1 - t0) / N * 1e6


# calibration of torch.cuda._sleep's unit (u2pl_amd.comm.U2PL_EMULATE_COLL_US)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000); torch.cuda.synchronize()      # (the first launch loads the kernel: not part of the unit)
e0.record(); torch.cuda._sleep(1_000_000); e1.record(); torch.cuda.synchronize()
print(f"torch.cuda._sleep(1e6) = {e0.elapsed_time(e1) * 1e3:.0f} us -> {1e6 / (e0.elapsed_time(e1) * 1e3):.1f} ticks per us")
s = torch.cuda.Stream()
for mode in ("none", "sync", "async", "hop", "none", "sync"):
    run(mode)
    g, h = run(mode)
    print(f"{mode:6s}: GPU {g:6.1f} us per link, host {h:6.1f} us per link")
dist.destroy_process_group()
