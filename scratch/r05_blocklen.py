"""The launch-form decision table at the block lengths a host really uses (round-4 review item 6): the reference's run(handle, n_samples)
is called with the host's period — 64 or 128 frames on MOD devices and low-latency desktop set-ups (rt-neural-generic.cpp:484) — while
every crossover in many_streams_form / use_pipe / split_pays was measured at 256-frame blocks. For cfg2's, cfg3's and the LSTM-64 model at
256 ... 16 384 streams and 64 / 128 / 256 frames: the pool's own pick against every form that can be forced, per-block time from HIP events.

    python scratch/r05_blocklen.py [frames,frames,...]  > profiles/r05_blocklen_forms.txt      (hooks build: it forces forms)"""
import importlib
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("AIDAX_LIB", os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "hooks", "libaidax_hip.so"))
import torch  # noqa: E402

ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
SWITCHES = ("AIDAX_KERNEL", "AIDAX_LS1", "AIDAX_GRU_GM", "AIDAX_LSTM_GS", "AIDAX_MFMA_LP")
MODELS = {
    "lstm32/1": (dict(kind="lstm", hidden=32, input_size=1, seed=32), {}),
    "gru64/3": (dict(kind="gru", hidden=64, input_size=3, seed=64), dict(bass_boost_db=4.0, mid_boost_db=-3.0, mid_q=1.2, treble_boost_db=2.0,
                                                                          depth_boost_db=3.0, presence_boost_db=3.0, param1=0.5, param2=0.3)),
    "lstm64/1": (dict(kind="lstm", hidden=64, input_size=1, seed=64), {}),
}
FORMS = {
    "lstm32/1": [("auto", {}), ("pipe", {"AIDAX_KERNEL": "pipe"}), ("split", {"AIDAX_KERNEL": "split"}), ("quad", {"AIDAX_KERNEL": "quad"}),
                 ("mfma", {"AIDAX_KERNEL": "mfma", "AIDAX_LS1": "0"}), ("ls1", {"AIDAX_KERNEL": "mfma", "AIDAX_LS1": "1"})],
    "gru64/3": [("auto", {}), ("split", {"AIDAX_KERNEL": "split"}), ("quad", {"AIDAX_KERNEL": "quad"}), ("gs", {"AIDAX_KERNEL": "mfma"}),
                ("gm-f32", {"AIDAX_KERNEL": "mfma", "AIDAX_GRU_GM": "f32"})],
    "lstm64/1": [("auto", {}), ("split", {"AIDAX_KERNEL": "split"}), ("quad", {"AIDAX_KERNEL": "quad"}), ("mfma", {"AIDAX_KERNEL": "mfma", "AIDAX_LS1": "0", "AIDAX_LSTM_GS": "0"}),
                 ("ls1", {"AIDAX_KERNEL": "mfma", "AIDAX_LS1": "1", "AIDAX_LSTM_GS": "0"}), ("lgs", {"AIDAX_KERNEL": "mfma", "AIDAX_LSTM_GS": "1"})],
}
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
paths = {}


def run(env, name, S, n):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    kw, ctl = MODELS[name]
    if name not in paths:
        paths[name] = W.write_model(W.make_model(**kw), os.path.join(tempfile.mkdtemp(), "m.json"))
    pool = ax.Pool(S, n)
    pool.set_model(ax.Model(paths[name]))
    pool.set_controls(ax.default_controls(**ctl))
    x = torch.rand(S, n, device="cuda") - 0.5
    y = torch.empty_like(x)
    t0 = time.time()
    it = 0
    while time.time() - t0 < 0.12:
        for _ in range(8):
            pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
            it += 8
        torch.cuda.synchronize()
    steps = max(20, min(400, it // 3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(steps):
        pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    kname = pool.kernel_name
    pool.close()
    return kname, e0.elapsed_time(e1) / steps * 1e3


frames = [int(f) for f in sys.argv[1].split(",")] if len(sys.argv) > 1 else [64, 128, 256]
streams = [int(s) for s in os.environ.get("BL_STREAMS", "256,1024,2048,4096,8192,16384").split(",")]
if os.environ.get("BL_MODELS"):
    MODELS = {k: v for k, v in MODELS.items() if k in os.environ["BL_MODELS"].split(",")}
print("us per block; * = the pool's own pick; [x.xx] = pick / best forced form of the row (1.00: the pick is the best)")
for name in MODELS:
    for n in frames:
        for S in streams:
            cells, times = [], {}
            for tag, env in FORMS[name]:
                try:
                    k, us = run(env, name, S, n)
                except Exception as e:       # a form that does not serve this shape / size
                    cells.append(f"{tag}: -")
                    continue
                times[tag] = (k, us)
                cells.append(f"{tag}={k}: {us:7.1f}")
            auto_k, auto_us = times["auto"]
            best_tag = min((t for t in times if t != "auto"), key=lambda t: times[t][1])
            ratio = auto_us / times[best_tag][1]
            print(f"{name:9s} n={n:3d} S={S:6d}  pick {auto_k:22s} {auto_us:7.1f}  best {best_tag:6s} {times[best_tag][1]:7.1f} [{ratio:4.2f}]  | " + " | ".join(cells[1:]), flush=True)
