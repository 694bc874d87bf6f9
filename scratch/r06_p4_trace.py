"""stations of ONE launch of k_lstm_pipe4<32> (measurement build -DAIDAX_P4_TRACE: scratch/prev_lib/p4trace), s_memrealtime (100 MHz) of recurrent
wave 0 and the helper wave of the workgroups of streams 0, 400 and 1020, relative to the earliest entry among them.
usage: AIDAX_LIB=scratch/prev_lib/p4trace/libaidax_hip.so python scratch/r06_p4_trace.py [frames]"""
import importlib, os, sys, tempfile, time, torch
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ax = importlib.import_module("aidadsp-lv2_amd"); W = ax.workloads
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = 1024
p = W.write_model(W.make_model("lstm", 32, 1, seed=32), os.path.join(tempfile.mkdtemp(), "m.json"))
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls(eq_bypass=1.0) if os.environ.get("EQ") == "0" else ax.default_controls())
x = torch.rand(S, n, device="cuda") - 0.5; y = torch.empty_like(x)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
for _ in range(200): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
torch.cuda.synchronize()
names_n = ["entry", "weights asked for", "past the barrier", "tile 0 ready", "tile 0 done, tile 1 ready", "last tile begins", "last tile done", "state stored"]
names_h = ["entry", "loads asked for", "past the barrier", "tick 0 done", "last tick begins", "last tick done", "rows stored (issued)", "stores landed"]
acc = []
tiles = []
for rep in range(20):
    for _ in range(3): pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    torch.cuda.synchronize()
    out = y.cpu().numpy()
    rows = []
    for s0 in (0, 400, 1020):
        t = out[s0, :32].copy().view(np.uint64)
        rows.append(t)
        if s0 == 0 and n >= 128:
            tiles.append(out[0, 32:32 + 2 * min(n // 16, 32)].copy().view(np.uint64).astype(np.int64))
    acc.append(np.stack(rows))
acc = np.stack(acc).astype(np.int64)            # [rep][wg][16]
base = acc[:, :, [0, 8]].min(axis=(1, 2), keepdims=True)
rel = (acc - base) / 100.0                      # us
med = np.median(rel, axis=0)
print(pool.kernel_name, n, "frames; us since the earliest entry (median of 20 launches); workgroups of streams 0 / 400 / 1020")
for k in range(8): print(f"  recurrent wave 0: {names_n[k]:28s} " + "  ".join(f"{med[w, k]:7.2f}" for w in range(3)))
for k in range(8): print(f"  helper wave:      {names_h[k]:28s} " + "  ".join(f"{med[w, 8 + k]:7.2f}" for w in range(3)))
if tiles:
    tl = np.median(np.diff(np.stack(tiles), axis=1), axis=0) / 100.0
    print("  recurrent wave 0 of stream 0, tile by tile (us from one tile's beginning to the next's):", " ".join(f"{v:.2f}" for v in tl))
