"""HW-queue assignment of the library's streams with / without an RCCL process group (AMD_LOG_LEVEL=3 in the env)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.zeros(8, device="cuda:0"); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "dist":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
sys.stderr.write("==== PLUGIN ====\n"); sys.stderr.flush()
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(100_000, 2))
v = headless_view(0); s = CloudSettings()
p.set_async(True); p.set_profiling(0)
p.set_pipeline_depth(8); p.set_pipeline_streams(4)
pv = p.prepare(v, s)
for _ in range(24): p.render(h, pv, download=False)
p.synchronize()
sys.stderr.write("==== DONE ====\n"); sys.stderr.flush()
