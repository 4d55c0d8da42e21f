"""Two ranks, gloo all-reduce of a 1080p CUDA buffer with no rendering at all (developer tool:
separates collective-backend stalls from this library's kernels)."""
import os, sys, time
import torch, torch.distributed as dist
rank = int(os.environ["RANK"])
def say(msg): print(f"[rank {rank}] {time.time():.2f} {msg}", flush=True)
torch.cuda.set_device(0)
dist.init_process_group("gloo")
x = torch.ones((1920, 1080, 3), dtype=torch.float32, device="cuda:0")
for it in range(5):
    t0 = time.time(); dist.all_reduce(x); torch.cuda.synchronize(); say(f"all_reduce {it}: {time.time() - t0:.3f} s")
    dist.barrier()
dist.destroy_process_group()
