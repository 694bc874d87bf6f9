"""hipLaunchCooperativeKernel for the chained kernels (AIDAX_LP_COOP=1) against the plain launch: cost per block on cfg5, and what
happens while ANOTHER PROCESS keeps the CUs busy (a child process launching long full-chip kernels back to back).
usage: python scratch/coop_ab.py"""
import importlib, os, subprocess, sys, tempfile, time
CHILD = r'''
import importlib, os, sys, tempfile, time, torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
p = W.write_model(W.make_model("lstm", 96, 1, seed=96, n_rnn=2), os.path.join(tempfile.mkdtemp(), "m.json"))
S, n = 2048, 256
pool = ax.Pool(S, n); pool.set_model(ax.Model(p))
x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
t0 = time.time()
while time.time() - t0 < 0.3:
    for _ in range(8): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
steps = int(sys.argv[1])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record(st)
for _ in range(steps): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps * 1e6
err = "none"
try:
    pool.sync()
except Exception as e:
    err = str(e)[:120]
print("coop=%s %-14s %8.1f us per block (events) %8.1f us (host wall)  kernel now %-16s give-up: %s" % (os.environ.get("AIDAX_LP_COOP", "0"), sys.argv[2], e0.elapsed_time(e1) / steps * 1e3, wall, pool.kernel_name, err), flush=True)
'''
HOG = r'''
import time, torch
a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
t0 = time.time()
while time.time() - t0 < float(__import__("sys").argv[1]):
    for _ in range(4): c = a @ b
    torch.cuda.synchronize()
'''
def run(coop, tag, steps, hog=False):
    env = dict(os.environ, AIDAX_LP_COOP=coop)
    h = None
    if hog:
        h = subprocess.Popen([sys.executable, "-c", HOG, "25"])
        time.sleep(6)
    r = subprocess.run([sys.executable, "-c", CHILD, str(steps), tag], env=env, capture_output=True, text=True, timeout=300)
    print((r.stdout.strip() or r.stderr.strip()[-300:]), flush=True)
    if h: h.wait()
for coop in ("0", "1"):
    run(coop, "alone", 300)
for coop in ("0", "1"):
    run(coop, "next to a hog", 100, hog=True)
