# device / host memory before and after many pool and hub life cycles with model swaps (leak check)
import importlib, os, sys, tempfile, resource
import torch
sys.path.insert(0, os.getcwd())
import numpy as np
from tests import modelgen
ax = importlib.import_module("aidadsp-lv2_amd")
d = tempfile.mkdtemp()
models = [ax.Model(modelgen.write_model(modelgen.make_model(**kw), os.path.join(d, f"m{i}.json"))) for i, kw in enumerate(
    [dict(kind="lstm", hidden=16, input_size=1, seed=1), dict(kind="lstm", hidden=32, input_size=1, seed=3, n_rnn=2),
     dict(kind="conv", hidden=16, input_size=1, seed=4), dict(kind="gru", hidden=64, input_size=3, seed=5)])]
def cycle():
    pool = ax.Pool(64, 256)
    x = np.zeros((64, 256), np.float32)
    for m in models:
        pool.set_model(m); pool.process(x)
        sg = pool.prepare_model(models[0]); pool.commit_model(sg); pool.staged_free(sg); pool.process(x)
    pool.close()
    hub = ax.Hub(8, 128); hub.set_deadline_us(0); hub.set_model(models[1])
    s = [hub.attach() for _ in range(4)]
    for _ in range(3):
        for q in s: hub.run(q, np.zeros(128, np.float32))
    hub.set_model(models[2]); hub.detach(s[0]); hub.flush(); hub.close()
for _ in range(5): cycle()
torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]; rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for _ in range(60): cycle()
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]; rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print("device free before/after 60 cycles:", free0, free1, "delta MiB", (free0 - free1) / 2**20, " host maxrss delta MiB", (rss1 - rss0) / 1024)
