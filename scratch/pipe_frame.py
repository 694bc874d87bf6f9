"""Where one frame of cfg2's recurrent wave goes: s_memtime at six points of frame 8 of the unrolled stage (LstmCell<32>::step, PT_STAMP) and after the
frame behind it. Measurement build:  make -s -j8 OBJDIR=build/obj_pt LIBDIR=build/lib_pt EXTRA=-DAIDAX_PIPE_TRACE build/lib_pt/libaidax_hip.so
AIDAX_LIB=build/lib_pt/libaidax_hip.so python scratch/pipe_frame.py"""
import importlib, os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
j = modelgen.make_model("lstm", 32, 1, seed=32); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
S = 1024
pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
for _ in range(300): pool.process_device(x.data_ptr(), y.data_ptr(), 256)
torch.cuda.synchronize()
print(pool.kernel_name)
raw = y[:, :56].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
cyc = raw[:, 2]
t = raw[:, 8:15]
names = ["frame starts (h reads issued next)", "own unit + 30 rotations issued, the other row's h arrived (lgkmcnt 0)", "16 packed FMAs + 3 adds",
         "gate activations issued", "c, tanh(c), h(t) in registers", "h published (ds_write issued)", "the NEXT frame completely issued"]
print(f"workgroup: {np.median(cyc):.0f} cycles of s_memtime start -> end (median), {np.median(cyc) / 256:.1f} per frame")
prev = np.zeros(S)
for k in range(1, 7):
    d = t[:, k] - t[:, k - 1]
    print(f"  {names[k]:75s} +{np.median(d):6.0f}  (p10 {np.percentile(d, 10):5.0f}, p90 {np.percentile(d, 90):5.0f})   at {np.median(t[:, k]):6.0f}")
print(f"frame 8 start -> frame 9 issued: {np.median(t[:, 6]):.0f} cycles; frame 8 alone (start -> h published): {np.median(t[:, 5]):.0f}")

st = raw[:, 16:53].astype(np.int64)
b, e = st[:, 0:36:2], st[:, 1:36:2]                      # begin / end of wave N's part of pipeline step p (after / before the step's barrier)
print(f"wave N enters the pipeline loop at {np.median(b[:, 0]):.0f} cycles (prologue: weights, state, h(-1)); leaves it at {np.median(st[:, 36]):.0f}; workgroup ends at {np.median(cyc):.0f}")
work = e - b
gap = b[:, 1:] - e[:, :-1]                               # the barrier between step p and p + 1, as wave N sees it
print("step:  work of wave N | barrier behind it   (cycles, medians)")
for p_ in range(18):
    print(f"  {p_:2d}   {np.median(work[:, p_]):7.0f} | {np.median(gap[:, p_]) if p_ < 17 else 0:7.0f}")
print(f"steps 1..16: work {np.median(work[:, 1:17].sum(axis=1)):.0f} = {np.median(work[:, 1:17].sum(axis=1)) / 256:.1f} per frame; barriers {np.median(gap[:, 1:17].sum(axis=1)):.0f}")
