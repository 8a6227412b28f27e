"""the fresh-interpreter test (product initialises HIP, THEN `import torch`) once did not return within 600 s.
Stress it: N children in a row, each armed with faulthandler (a stack after 120 s, then exit), wall time per child; then load a
DIFFERENT build of libnanocall_hip.so in this process (the condition the hang was seen under: an A/B session that had swapped
kernel builds), and N more.   python tools/order_child_stress.py [N=50] > profiles/r04_order_child_stress.txt"""
import ctypes, os, shutil, subprocess, sys, tempfile, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
src = open(os.path.join(ROOT, "tests", "test_viterbi_gpu.py")).read()
child = src[src.index('_ORDER_CHILD = r"""') + len('_ORDER_CHILD = r"""'):]
child = child[:child.index('"""')]
tmp = tempfile.mkdtemp()
script = os.path.join(tmp, "order_child.py")
open(script, "w").write(child)


def batch(label):
    walls, bad = [], 0
    for i in range(N):
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, script, ROOT], capture_output=True, text=True, timeout=200)
            ok = p.returncode == 0 and p.stdout.strip().startswith("ok")
            tail = (p.stdout.strip().splitlines() or [""])[-1]
            if not ok:
                tail = (p.stdout[-500:] + " || " + p.stderr[-3000:]).replace("\n", " | ")
        except subprocess.TimeoutExpired as e:
            ok, tail = False, "TIMEOUT (200 s; faulthandler should have fired at 120 s): " + repr((e.stderr or b"")[-3000:])
        dt = time.time() - t0
        walls.append(dt)
        bad += not ok
        if not ok or i < 2 or dt > 30:
            print(f"[{label} {i:3d}] {'ok ' if ok else 'BAD'} {dt:6.1f} s  {tail}", flush=True)
    print(f"== {label}: {N - bad} / {N} ok; wall per child min {min(walls):.1f} s, median {sorted(walls)[N // 2]:.1f} s, max {max(walls):.1f} s", flush=True)
    return bad


bad = batch("fresh box")
# a different build of the library (another symbol table / code object: rebuilt with an extra -D), loaded and used in THIS process
alt = os.path.join(tmp, "alt")
shutil.copytree(os.path.join(ROOT, "nanocall_amd"), os.path.join(alt, "nanocall_amd"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join(alt, "include"))
r = subprocess.run("make -s -C %s clean && make -s -C %s -j16 CXXFLAGS='-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. -Wall -Wno-unused-result -DNCHMM_STRESS_ALT_BUILD=1'"
                   % (os.path.join(alt, "nanocall_amd", "csrc"), os.path.join(alt, "nanocall_amd", "csrc")), shell=True, capture_output=True, text=True)
print("alt build rc", r.returncode, r.stderr[-300:].replace("\n", " | "), flush=True)
if r.returncode == 0:
    run_alt = ("import sys; sys.path.insert(0, %r); import numpy as np; import nanocall_amd as na; from nanocall_amd import synth; t = na.builtin_model('r73.t');"
               "ev = synth.generate(t, 8, 400); off, m, s, st = synth.flat_batch(ev); cm, sd, ls = na.events_prepare(m, s, st, 0.0); ctx = na.Context(0);"
               "ctx.put_model(0, na.scaled_model_table(t)); ctx.put_transitions(0, *na.transitions_fast(0.3, 0.1)); print(ctx.viterbi(off, cm, sd, ls)[1][:2], na._lib.LIB_PATH)") % alt
    p = subprocess.run([sys.executable, "-c", run_alt], capture_output=True, text=True, timeout=600)
    print("alt library run:", p.stdout.strip()[-200:], p.stderr.strip()[-200:].replace("\n", " | "), flush=True)
bad += batch("after another build was loaded")
print("TOTAL BAD", bad)
