"""Rehearsal of the RCCL code paths on a 1-GPU box: a 1-rank `nccl` process group drives ShardedRollout (barrier, max / sum
all-reduce on device tensors) exactly as bench.py does for N > 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
import torch, torch.distributed as dist
import importlib
import __graft_entry__ as ge
ge.package()
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
mod = importlib.import_module("drl-tetris_amd.distributed")
sh = mod.ShardedRollout(65536, 1, 20, 10, rank=0, world=1, device=0, dist=dist)
sh.run(64, 1)
res = sh.run(1024, 1)
print("nccl sharded rollout:", res["counters"].tolist(), "%.2f us/launch" % (res["event_ms"] * 1e3 / 1024))
sh.close()
dist.destroy_process_group()
print("ok")
