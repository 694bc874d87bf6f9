"""k_*_pipe4 against k_*_pipe for every cell of the reference's table that has both (hooks build: AIDAX_PIPE4=0 forces the three-wave pipeline),
1024 streams x 256-frame blocks, kernel time from HIP events; one child process per (cell, form) so that the switch is read afresh.
usage: [INPUTS=2|3] [SMALL=1] python scratch/r06_pipe4_cells.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import importlib, os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd"); W = ax.workloads
kind, H, S, n, I = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
p = W.write_model(W.make_model(kind, H, I, seed=H), os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(param1=0.6, param2=0.3))
x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
import time
t0 = time.time()
while time.time() - t0 < 0.25:
    for _ in range(16): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 600
e0.record(st)
for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
e1.record(st); torch.cuda.synchronize()
print(pool.kernel_name, round(e0.elapsed_time(e1) / N * 1e3, 2))
'''
env0 = dict(os.environ, AIDAX_LIB=os.environ.get("CELLS_LIB") or os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so"))      # CELLS_LIB: a measurement build (-DAIDAX_P4_ALL_CELLS: every cell up to 32 has the kernel)
S = int(os.environ.get("STREAMS", "1024"))
I = os.environ.get("INPUTS", "1")          # 2 / 3: conditioned models (PARAM1 / PARAM2 as model inputs)
SMALL = os.environ.get("SMALL") == "1"     # only the cells k_*_pipe4 is built for
MID = os.environ.get("MID") == "1"         # only the cells whose plain models keep k_*_pipe
print(f"# {S} streams, input_size {I}, us per block; k_*_pipe (AIDAX_PIPE4=0) | k_*_pipe4 (default)   at 256 / 64 frames")
for kind, hs in ((("lstm", (20, 24)), ("gru", (20, 24, 32))) if MID else (("lstm", (8, 12, 16, 32)), ("gru", (8, 12, 16))) if SMALL else (("lstm", (8, 12, 16, 20, 24, 32)), ("gru", (8, 12, 16, 20, 24, 32))) if I != "1" else (("lstm", (8, 12, 16, 20, 24, 32, 40)), ("gru", (8, 12, 16, 20, 24, 32, 40, 64)))):
    for H in hs:
        row = []
        for n in (256, 64):
            for p4 in ("0", "1"):
                env = dict(env0, AIDAX_PIPE4=p4)
                r = subprocess.run([sys.executable, "-c", CHILD, kind, str(H), str(S), str(n), I], env=env, cwd=ROOT, capture_output=True, text=True)
                row.append(r.stdout.strip() or r.stderr.strip()[-120:])
        print(f"{kind}-{H:<3d} 256: {row[0]:28s} | {row[1]:28s}   64: {row[2]:28s} | {row[3]:28s}", flush=True)
