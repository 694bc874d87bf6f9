# k_mfma_lp under stress: full-size pools (every CU busy, 128 stream groups x 2 layers), many launches of random
# length, against k_mfma on the same inputs. State must match bit for bit after every launch, outputs to 5e-7.
import importlib, os, sys, tempfile
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
ax = importlib.import_module("aidadsp-lv2_amd")
W = ax.workloads
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(7)
for H, NL, S in ((96, 2, 2048), (64, 2, 2048), (48, 3, 1344), (32, 2, 2048)):
    p = W.write_model(W.make_model("lstm" if H != 48 else "gru", H, 1, seed=H, n_rnn=NL), os.path.join(tempfile.mkdtemp(), "m.json"))
    sizes = [int(rs.choice([256, 256, 256, 1, 7, 64, 255, 257, 512, 1000, 31])) for _ in range(launches)]
    res = {}
    for lp in ("1", "0"):
        os.environ["AIDAX_MFMA_LP"] = lp
        pool = ax.Pool(S, 1024); pool.set_model(ax.Model(p)); pool.set_controls(ax.default_controls())
        st = torch.cuda.Stream(); torch.cuda.set_stream(st)
        g = torch.Generator(device="cuda"); g.manual_seed(11)
        outs, states = [], []
        for i, n in enumerate(sizes):
            x = (torch.rand(S, n, device="cuda", generator=g) - 0.5).contiguous(); y = torch.empty_like(x)
            pool.process_device(x.data_ptr(), y.data_ptr(), n, st.cuda_stream)
            if i % 20 == 19 or i == len(sizes) - 1:
                torch.cuda.synchronize()
                outs.append(y.cpu().numpy().copy())
                states.append([pool.read_state(stream=s, layer=NL - 1, hidden=128)[0].copy() for s in (0, S // 2 + 3, S - 1)])
        pool.sync()
        res[lp] = (outs, states, pool.kernel_name); pool.close()
    worst = max(float(np.abs(a - b).max()) for a, b in zip(res["1"][0], res["0"][0]))
    same = all(np.array_equal(a, b) for sa, sb in zip(res["1"][1], res["0"][1]) for a, b in zip(sa, sb))
    print(f"H={H} x{NL} S={S}: {launches} launches, {res['1'][2]} vs {res['0'][2]}: state identical={same}, worst output diff={worst:.3e}", flush=True)
    assert same and worst < 5e-7
print("lp stress ok")
