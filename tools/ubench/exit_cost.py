import subprocess, sys, time, os
def run(gb):
    code = f"import torch, os, time; x = torch.empty(int({gb}*2**30), dtype=torch.uint8, device='cuda') if {gb} else None; torch.cuda.synchronize(); print(time.time(), flush=True); os._exit(0)"
    t0 = time.time(); p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True); t1 = time.time()
    t_exit_begin = float(p.stdout.strip().splitlines()[-1])
    return round(t1 - t_exit_begin, 3), round(t1 - t0, 3)
for gb in (0, 1, 8, 24, 48):
    print(gb, "GiB: exit takes", *run(gb))
