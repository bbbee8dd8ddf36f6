"""Does a ONE-rank nccl (= RCCL) process group work on this box?  Prints each stage; dumps the stack if a stage hangs."""
import faulthandler, os, sys, time
faulthandler.dump_traceback_later(60, exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
print("init ...", flush=True)
t0 = time.time()
if len(sys.argv) > 1 and sys.argv[1] == "device_id":
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("nccl")
print("init done", time.time() - t0, flush=True)
x = torch.arange(8, dtype=torch.float64, device="cuda:0")
out = [torch.empty_like(x)]
dist.all_gather(out, x); torch.cuda.synchronize(); print("all_gather ok", out[0].tolist(), flush=True)
dist.barrier(); torch.cuda.synchronize(); print("barrier ok", flush=True)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda:0"); dist.all_reduce(t, op=dist.ReduceOp.MAX); print("all_reduce ok", t.item(), flush=True)
dist.destroy_process_group(); print("done", flush=True)
