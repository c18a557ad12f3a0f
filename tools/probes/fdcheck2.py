import os
os.environ["HIP_VISIBLE_DEVICES"] = ""; os.environ["CUDA_VISIBLE_DEVICES"] = ""
def fds():
    out=[]
    for f in os.listdir("/proc/self/fd"):
        try: out.append(os.readlink("/proc/self/fd/"+f))
        except OSError: pass
    return [x for x in out if "kfd" in x or "dri" in x]
import torch, torch.distributed as dist
from datetime import timedelta
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
print("before init", fds())
dist.init_process_group("gloo", timeout=timedelta(seconds=60))
print("after init_process_group(gloo)", fds())
t = torch.zeros(1)
dist.all_reduce(t)
print("after all_reduce", fds())
dist.barrier()
print("after barrier", fds())
objs = [None]
dist.all_gather_object(objs, {"a": 1})
print("after all_gather_object", fds())
dist.destroy_process_group()
