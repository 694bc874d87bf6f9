import importlib, os, sys, tempfile, torch
sys.path.insert(0, os.getcwd())
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
for kind, H in (("lstm", 12), ("lstm", 16), ("lstm", 32), ("gru", 16)):
    j = modelgen.make_model(kind, H, 1, seed=H); p = modelgen.write_model(j, os.path.join(tempfile.mkdtemp(), "m.json"))
    for S in (1, 256, 1024):
        pool = ax.Pool(S, 256); pool.set_model(ax.Model(p))
        x = torch.rand(S, 256, device="cuda") - 0.5; y = torch.empty_like(x)
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        for _ in range(300): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        N = 2000
        e0.record()
        for _ in range(N): pool.process_device(x.data_ptr(), y.data_ptr(), 256, st.cuda_stream)
        e1.record(); torch.cuda.synchronize()
        print(kind, H, S, pool.kernel_name, round(e0.elapsed_time(e1) / N * 1e3, 2), "us per pass")
        pool.close()
