"""tools/make_bundle.py: the generated bundle describes the same ports the shell
implements, with the defaults the C ABI uses, and ships the default model its state block points at."""
import importlib
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_bundle  # noqa: E402
import make_ports_json  # noqa: E402

ax = importlib.import_module("aidadsp-lv2_amd")


def _shell_port_enum():
    src = open(os.path.join(ROOT, "aidadsp-lv2_amd", "lv2", "rt_neural_generic_lv2.cpp")).read()
    body = re.search(r"enum PortIndex \{[^\n]*\n(.*?)\};", src, re.S).group(1)
    names = [t.split("=")[0].strip() for t in body.replace("\n", " ").split(",") if t.strip()]
    return [n for n in names if n != "PLUGIN_PORT_COUNT"]


def test_generated_ttl_matches_shell_and_abi_defaults(tmp_path):
    bundle = make_bundle.make_bundle(str(tmp_path), binaries=False)
    ttl = open(os.path.join(bundle, "rt-neural-generic.ttl")).read()
    ports = re.findall(r"lv2:index (\d+) ;\s*lv2:symbol \"([^\"]+)\"", ttl)
    assert [int(i) for i, _ in ports] == list(range(26))
    enum = _shell_port_enum()
    assert len(enum) == 26 and enum[25] == "LATENCY" and ports[25][1] == "latency"
    # spot anchors between the shell's enum and the symbols hosts see
    sym = [s for _, s in ports]
    for name, symbol in (("IN", "IN"), ("OUT_1", "OUT"), ("PLUGIN_CONTROL", "CONTROL"), ("PLUGIN_NOTIFY", "NOTIFY"),
                         ("IN_LPF", "ANTIALIASING"), ("NET_BYPASS", "NETBYPASS"), ("EQ_POS", "EQPOS"), ("MIDQ", "MIDQ"),
                         ("MASTER", "MASTER"), ("INPUT_SIZE", "ModelInSize"), ("PLUGIN_ENABLED", "enabled")):
        assert sym[enum.index(name)] == symbol
    # defaults of the control inputs == aidax_controls_default (rt-neural-generic.ttl:94-313)
    c = ax.default_controls()
    defaults = dict(re.findall(r"lv2:symbol \"([^\"]+)\" ;\s*lv2:name \"[^\"]+\" ;\s*lv2:default ([-0-9.e]+)", ttl))
    fields = ax.binding.CONTROL_FIELDS
    assert len(make_bundle.CONTROL_SYMBOLS) == len(fields) == 20
    for symbol, field in zip(make_bundle.CONTROL_SYMBOLS, fields):
        assert abs(float(defaults[symbol]) - getattr(c, field)) < 1e-5, (symbol, field)     # float32 fields
    assert "work:schedule" in ttl and "state:interface" in ttl and "patch:writable" in ttl


def test_bundle_layout(tmp_path):
    bundle = make_bundle.make_bundle(str(tmp_path), binaries=False)
    man = open(os.path.join(bundle, "manifest.ttl")).read()
    assert "<rt-neural-generic.so>" in man and make_bundle.URI in man
    assert os.path.exists(os.path.join(bundle, make_bundle.DEFAULT_MODEL))
    assert len(os.listdir(os.path.join(bundle, "models", "deer ink studios"))) == 6
    m = ax.Model(os.path.join(bundle, make_bundle.DEFAULT_MODEL))          # loads through the C ABI
    assert (m.info.cell, m.info.hidden, m.info.input_size) == (0, 12, 1)


def test_generated_ttl_carries_the_reference_port_table(tmp_path):
    """tests/golden/ports.json is the port table of the REFERENCE's rt-neural-generic.ttl:61-317 (extracted by
    tests/golden/make_ports_json.py). The generated description, read back through the same parser, must carry
    exactly those 25 ports (index, symbol, name, types, default, range, unit, properties, designation, scale
    points), the same required features / extension data and the same default state; the one extra port sits
    behind them."""
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "ports.json")))
    bundle = make_bundle.make_bundle(str(tmp_path), binaries=False)
    got = make_ports_json.parse_ttl(open(os.path.join(bundle, "rt-neural-generic.ttl")).read())
    assert got["plugin_uri"] == want["plugin_uri"]
    assert got["required_features"] == want["required_features"]
    assert got["extension_data"] == want["extension_data"]
    assert got["default_state"] == want["default_state"]
    assert len(want["ports"]) == 25 and len(got["ports"]) == 26
    for g, w in zip(got["ports"], want["ports"]):
        assert g == w, (w["index"], g, w)
    extra = got["ports"][25]
    assert extra["symbol"] == "latency" and "reportsLatency" in extra["properties"] and extra["designation"] == "latency"
    assert "lv2:OutputPort" in extra["types"]


@pytest.mark.gpu
def test_plugin_loaded_from_the_built_bundle_like_a_host_would(tmp_path):
    """`make bundle` equivalent into a scratch dir; a fresh process (cwd elsewhere, no LD_LIBRARY_PATH) loads
    rt-neural-generic.so from the bundle directory as named by manifest.ttl, restores the TTL's default state
    relative to the bundle and plays: audio == oracle, and the HIP library that got mapped is the bundle's own copy
    (rpath $ORIGIN), not the in-tree build."""
    import subprocess
    bundle = make_bundle.make_bundle(str(tmp_path), binaries=True)
    dyn = subprocess.run(["readelf", "-d", os.path.join(bundle, "rt-neural-generic.so")], capture_output=True, text=True).stdout
    assert "$ORIGIN" in dyn and "libaidax_hip.so" in dyn
    env = {k: v for k, v in os.environ.items() if k not in ("LD_LIBRARY_PATH", "LD_PRELOAD")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bundle_host.py"), bundle], cwd=str(tmp_path), env=env,
                         capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-3000:]
    out = json.loads(run.stdout.strip().splitlines()[-1])
    assert out["max_abs_err"] < 1.0e-5 and out["peak"] > 1e-3, out
    assert out["state"] == make_bundle.DEFAULT_MODEL and out["saved"] == [make_bundle.DEFAULT_MODEL]
    assert out["model_in_size"] == 1.0 and out["latency"] == 0.0       # one-stream mode adds no latency
    libs = [m for m in out["maps"] if m.endswith("libaidax_hip.so")]
    assert libs == [os.path.join(bundle, "libaidax_hip.so")], out["maps"]
    assert os.path.join(bundle, "rt-neural-generic.so") in out["maps"]
