import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench, nanocall_amd as na, torch
ctx = na.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
r = bench.config3_leg(ctx, 16, int(os.environ.get("CPU_THREADS", 48)), True)
print(json.dumps(r["cpu_baseline"]))
