"""The torch.distributed calls of bench.py's N > 1 path and of distributed.py (reduce, gather, all_gather, all_reduce, barrier; backend nccl = RCCL, device_id at init) with a world of ONE on
the box's GPU: that the API is there in this torch build and that a ShardedRenderer's two assembles run through it.  (Two ranks cannot share a device under RCCL.)"""
import os, sys
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.distributed as dist, numpy as np
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
a = torch.arange(12, dtype=torch.float32, device="cuda").reshape(3, 4)
dist.reduce(a, dst=0, op=dist.ReduceOp.SUM); dist.all_reduce(a); dist.all_reduce(a, op=dist.ReduceOp.MAX)
out = [torch.zeros_like(a)]; dist.gather(a, out, dst=0); assert torch.equal(out[0], a)
out = [torch.zeros_like(a)]; dist.all_gather(out, a); assert torch.equal(out[0], a)
dist.barrier(); torch.cuda.synchronize()
import metal_raytracing_amd as mrt
from metal_raytracing_amd.distributed import gather_compact, reduce_accumulation
sc = mrt.CornellScene((64, 48)); r = mrt.Renderer((64, 48), sc, device=0); r.draw(3, wait=True)
buf = torch.zeros((48, 64, 4), dtype=torch.float32, device="cuda:0"); r.copy_accum_to(buf.data_ptr(), buf.numel() * 4); r.wait()
ref = buf.clone(); reduce_accumulation(buf, "tile", 0); assert torch.equal(buf, ref)
def pack(rank):
    t = torch.empty((r.shard_tiles(rank, 1) * 64, 4), dtype=torch.float32, device="cuda:0"); r.pack_owned_tiles(t.data_ptr(), t.numel() * 4); r.wait(); return t
gather_compact(buf, 0, pack=pack, unpack=lambda c, rk: None)
assert np.array_equal(buf.cpu().numpy(), r.accumulation())
dist.barrier(); dist.destroy_process_group(); r.close()
print("nccl world-1 ok:", torch.__version__, torch.cuda.get_device_name(0))
