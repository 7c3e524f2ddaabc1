import os, torch, torch.distributed as dist
torch.zeros(8, device="cuda:0"); torch.cuda.synchronize()
print("==== BEFORE INIT ====", flush=True)
import sys; sys.stderr.write("==== BEFORE INIT ====\n"); sys.stderr.flush()
env0 = dict(os.environ)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
sys.stderr.write("==== AFTER INIT ====\n"); sys.stderr.flush()
print({k: v for k, v in os.environ.items() if env0.get(k) != v})
