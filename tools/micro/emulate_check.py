"""Does U2PL_EMULATE_COLL_US put its spin on the caller's stream?  (world of one on RCCL; run with U2PL_DIST_SINGLE=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ["U2PL_EMULATE_COLL_US"] = sys.argv[1] if len(sys.argv) > 1 else "200"
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from u2pl_amd import comm
print("emulate", comm._EMULATE_US, "dist_active", comm.dist_active())
t = torch.zeros(1024, device="cuda", dtype=torch.float64)
for s in (torch.cuda.current_stream(), torch.cuda.Stream()):
    with torch.cuda.stream(s):
        for _ in range(10):
            comm._all_reduce(t, "syncbn_allreduce")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h0 = time.perf_counter()
        e0.record()
        for _ in range(200):
            comm._all_reduce(t, "syncbn_allreduce")
        e1.record()
        h1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"stream {s}: {e0.elapsed_time(e1) * 5:.1f} us per exchange on the device, {(h1 - h0) * 5e3:.1f} us host enqueue")
dist.destroy_process_group()
