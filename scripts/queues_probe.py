"""What does a process that has initialised RCCL have that makes 8 lanes on 4+ streams fast? (plain enqueue loop:
14.0 k fps at 8 lanes / 4 streams, 19.2 k after a one-rank process group + one all_reduce.)
python scripts/queues_probe.py <what>   what = plain | dist | dist_noop | dist_destroy | prio_stream | many_streams | spin"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
what = sys.argv[1] if len(sys.argv) > 1 else "plain"
if what.startswith("dist"):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    if what != "dist_noop":
        t = torch.zeros(8, device="cuda:0"); dist.all_reduce(t); torch.cuda.synchronize()
    if what == "dist_destroy":
        dist.destroy_process_group()
elif what == "prio_stream":
    s_hi = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s_hi):
        torch.zeros(8, device="cuda:0").add_(1)
    torch.cuda.synchronize()
elif what == "many_streams":
    ss = [torch.cuda.Stream() for _ in range(8)]
    for st in ss:
        with torch.cuda.stream(st):
            torch.zeros(8, device="cuda:0").add_(1)
    torch.cuda.synchronize()
elif what == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(ctypes.c_uint(1)))
elif what == "big_alloc":
    keep = torch.empty(6 << 30, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
elif what == "stack_limit":
    hip = ctypes.CDLL("libamdhip64.so")
    torch.zeros(8, device="cuda:0")
    print("hipDeviceSetLimit(stack, 16384) ->", hip.hipDeviceSetLimit(ctypes.c_int(0), ctypes.c_size_t(16384)))
elif what == "finegrained":
    hip = ctypes.CDLL("libamdhip64.so")
    torch.zeros(8, device="cuda:0")
    ptr = ctypes.c_void_p()
    print("hipExtMallocWithFlags(finegrained) ->", hip.hipExtMallocWithFlags(ctypes.byref(ptr), ctypes.c_size_t(64 << 20), ctypes.c_uint(1)))
    hp = ctypes.c_void_p()
    print("hipHostMalloc ->", hip.hipHostMalloc(ctypes.byref(hp), ctypes.c_size_t(64 << 20), ctypes.c_uint(0)))
elif what == "busy_thread":
    import threading
    stop = False
    def spin():
        while not stop:
            pass
    for _ in range(2):
        threading.Thread(target=spin, daemon=True).start()
elif what == "tensor":
    torch.zeros(8, device="cuda:0").add_(1); torch.cuda.synchronize()
from bevy_gaussian_splatting_amd import CloudSettings, GaussianSplattingPlugin, random_gaussians_3d_seeded
from bevy_gaussian_splatting_amd.multiview import headless_view
p = GaussianSplattingPlugin(0)
h = p.upload(random_gaussians_3d_seeded(1_000_000, 2))
v = headless_view(0); s = CloudSettings()
p.set_async(True); p.set_profiling(0)
for lanes, streams in ((8, 8), (8, 4), (6, 6), (8, 6)):
    p.set_pipeline_depth(lanes); p.set_pipeline_streams(streams)
    pv = p.prepare(v, s)
    for _ in range(60): p.render(h, pv, download=False)
    p.synchronize(); t0 = time.perf_counter()
    for _ in range(600): p.render(h, pv, download=False)
    p.synchronize(); dt = time.perf_counter() - t0
    print(f"{what} maxq={os.environ.get('GPU_MAX_HW_QUEUES')} lanes {lanes} streams {streams}: {600/dt:.0f} fps", flush=True)
