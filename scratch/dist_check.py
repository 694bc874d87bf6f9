# RCCL next to libaidax_hip.so in one process (world_size 1): the reduction path of bench.py
import importlib, os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
ax = importlib.import_module("aidadsp-lv2_amd")
import bench
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
pool = ax.Pool(64, 256)
x = torch.rand(64, 256, device="cuda"); y = torch.empty_like(x)
pool.set_loading(False)
pool.process_device(x.data_ptr(), y.data_ptr(), 256, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX)
n = torch.tensor([2.0], dtype=torch.float64, device="cuda"); dist.all_reduce(n, op=dist.ReduceOp.SUM)
res = ("rccl ok", float(t), float(n), float(y.abs().max()))
dist.destroy_process_group()
import ctypes
sys.stdout.flush(); ctypes.CDLL(None).fflush(None)
print(*res, flush=True)
