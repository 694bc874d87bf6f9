#!/usr/bin/env python3
"""Assembles an LV2 bundle around the MI355X build of rt-neural-generic:

    <out>/rt-neural-generic.lv2/
        manifest.ttl, rt-neural-generic.ttl     generated HERE from the port table below
        rt-neural-generic.so                    linked here with rpath $ORIGIN
        libaidax_hip.so                         copied from the build
        models/deer ink studios/*.json          the six bundled model files

The port table restates the plugin's public interface — index, symbol, range and default of every
port (rt-neural-generic/ttl/rt-neural-generic.ttl:61-313; enum rt-neural-generic.h:84-112) — so that a host
(jalv, mod-host) and saved sessions see the same ports as with the reference binary. The Turtle text
itself is written by this script, not copied.

    python tools/make_bundle.py [--out build] [--no-binaries]
"""
import argparse
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
URI = "http://aidadsp.cc/plugins/aidadsp-bundle/rt-neural-generic"
DEFAULT_MODEL = "models/deer ink studios/tw40_california_clean_deerinkstudios.json"

TOGGLE = ("integer", "toggled")
ENUM = ("integer", "enumeration")
# (symbol, name, kind, default, minimum, maximum, unit, properties, scale points)
PORTS = [
    ("IN", "IN", "audio_in", None, None, None, None, (), ()),
    ("OUT", "OUT", "audio_out", None, None, None, None, (), ()),
    ("CONTROL", "CONTROL", "atom_in", None, None, None, None, (), ()),
    ("NOTIFY", "NOTIFY", "atom_out", None, None, None, None, (), ()),
    ("ANTIALIASING", "ANTIALIASING", "control_in", 66.216, 0, 100.0, "pc", (), (("Off", 0),)),
    ("PREGAIN", "INPUT", "control_in", 0, -12.0, 12.0, "db", (), ()),
    ("NETBYPASS", "NETBYPASS", "control_in", 0, 0, 1, None, TOGGLE, ()),
    ("PARAM1", "PARAM1", "control_in", 0, 0, 1.0, None, (), ()),
    ("PARAM2", "PARAM2", "control_in", 0, 0, 1.0, None, (), ()),
    ("EQBYPASS", "EQBYPASS", "control_in", 0, 0, 1, None, TOGGLE, ()),
    ("EQPOS", "EQPOS", "control_in", 0, 0, 1, None, ENUM, (("POST", 0), ("PRE", 1))),
    ("BASS", "BASS", "control_in", 0, -8.0, 8, "db", (), ()),
    ("BFREQ", "BFREQ", "control_in", 305.0, 75.0, 600.0, "hz", (), ()),
    ("MID", "MID", "control_in", 0, -8.0, 8, "db", (), ()),
    ("MFREQ", "MFREQ", "control_in", 750.0, 150.0, 5000.0, "hz", (), ()),
    ("MIDQ", "MIDQ", "control_in", 0.707, 0.2, 5.0, None, (), ()),
    ("MTYPE", "MTYPE", "control_in", 0, 0, 1, None, ENUM, (("PEAK", 0), ("BANDPASS", 1))),
    ("TREBLE", "TREBLE", "control_in", 0, -8.0, 8, "db", (), ()),
    ("TFREQ", "TFREQ", "control_in", 2000.0, 1000.0, 4000.0, "hz", (), ()),
    ("DEPTH", "DEPTH", "control_in", 0, -8.0, 8, "db", (), ()),
    ("PRESENCE", "PRESENCE", "control_in", 0, -8.0, 8, "db", (), ()),
    ("DCBLOCKER", "DCBLOCKER", "control_in", 1, 0, 1, None, TOGGLE, ()),
    ("MASTER", "OUTPUT", "control_in", 0, -15.0, 15, "db", (), ()),
    ("ModelInSize", "Model Input Size", "control_out", 0, 0, 3, None, ENUM,
     (("ERROR", 0), ("SNAPSHOT", 1), ("WITH 1 PARAM", 2), ("WITH 2 PARAMS", 3))),
    ("enabled", "Enabled", "control_in", 1, 0, 1, None, (), ()),
]
# Not in the reference: hub mode delays every instance by one period, which a host must be told
# (lv2:reportsLatency). Appended AFTER the reference's 25 ports so that their indices stay what saved sessions
# and the reference's own TTL say; the shell tolerates hosts that never connect it.
EXTRA_PORTS = [
    ("latency", "Latency", "control_out", 0, 0, 8192, "frame", ("integer", "reportsLatency"), ()),
]
# the control inputs in the order of aidax_controls (include/aidax.h)
CONTROL_SYMBOLS = [p[0] for p in PORTS if p[2] == "control_in"]

PREFIXES = {
    "atom": "http://lv2plug.in/ns/ext/atom#", "doap": "http://usefulinc.com/ns/doap#",
    "lv2": "http://lv2plug.in/ns/lv2core#", "patch": "http://lv2plug.in/ns/ext/patch#",
    "rdf": "http://www.w3.org/1999/02/22-rdf-syntax-ns#", "rdfs": "http://www.w3.org/2000/01/rdf-schema#",
    "state": "http://lv2plug.in/ns/ext/state#", "urid": "http://lv2plug.in/ns/ext/urid#",
    "work": "http://lv2plug.in/ns/ext/worker#", "mod": "http://moddevices.com/ns/mod#",
    "units": "http://lv2plug.in/ns/extensions/units#",
}
KIND = {
    "audio_in": "lv2:AudioPort , lv2:InputPort", "audio_out": "lv2:AudioPort , lv2:OutputPort",
    "atom_in": "atom:AtomPort , lv2:InputPort", "atom_out": "atom:AtomPort , lv2:OutputPort",
    "control_in": "lv2:ControlPort , lv2:InputPort", "control_out": "lv2:ControlPort , lv2:OutputPort",
}


def _num(v):
    return repr(float(v)) if isinstance(v, float) else str(v)


def port_ttl(index, port):
    sym, name, kind, dflt, lo, hi, unit, props, points = port
    lines = [f"        a {KIND[kind]}", f"        lv2:index {index}", f'        lv2:symbol "{sym}"', f'        lv2:name "{name}"']
    if kind.startswith("atom"):
        lines += ["        atom:bufferType atom:Sequence", "        atom:supports patch:Message",
                  "        lv2:designation lv2:control"]
    if dflt is not None:
        lines += [f"        lv2:default {_num(dflt)}", f"        lv2:minimum {_num(lo)}", f"        lv2:maximum {_num(hi)}"]
    if unit:
        lines.append(f"        units:unit units:{unit}")
    for pr in props:
        lines.append(f"        lv2:portProperty lv2:{pr}")
    for label, value in points:
        lines.append(f'        lv2:scalePoint [ rdfs:label "{label}" ; rdf:value {value} ]')
    if sym == "enabled":
        lines.append("        lv2:designation lv2:enabled")
    if sym == "latency":
        lines.append("        lv2:designation lv2:latency")
    return "    [\n" + " ;\n".join(lines) + "\n    ]"


def plugin_ttl():
    head = "".join(f"@prefix {k}: <{v}> .\n" for k, v in PREFIXES.items())
    param = (f"\n<{URI}#json>\n    a lv2:Parameter ;\n    mod:fileTypes \"aidadspmodel\" ;\n"
             f"    rdfs:label \"Neural Model\" ;\n    rdfs:range atom:Path .\n")
    ports = " ,\n".join(port_ttl(i, p) for i, p in enumerate(PORTS + EXTRA_PORTS))
    model = DEFAULT_MODEL.replace(" ", "%20")
    body = (f"\n<{URI}>\n    a lv2:Plugin , lv2:SimulatorPlugin ;\n    doap:name \"AIDA-X (MI355X pool build)\" ;\n"
            "    doap:license <http://spdx.org/licenses/GPL-3.0-or-later.html> ;\n"
            "    lv2:minorVersion 1 ;\n    lv2:microVersion 1 ;\n"
            "    lv2:requiredFeature urid:map , work:schedule ;\n"
            "    lv2:optionalFeature lv2:hardRTCapable , state:loadDefaultState , state:mapPath ;\n"
            "    lv2:extensionData state:interface , work:interface ;\n"
            f"    patch:writable <{URI}#json> ;\n    lv2:port\n{ports} ;\n"
            f"    state:state [ <{URI}#json> <{model}> ] .\n")
    return head + param + body


def manifest_ttl():
    return (f"@prefix lv2: <{PREFIXES['lv2']}> .\n@prefix rdfs: <{PREFIXES['rdfs']}> .\n\n"
            f"<{URI}>\n    a lv2:Plugin ;\n    lv2:binary <rt-neural-generic.so> ;\n    rdfs:seeAlso <rt-neural-generic.ttl> .\n")


def make_bundle(out_dir, binaries=True):
    bundle = os.path.join(out_dir, "rt-neural-generic.lv2")
    os.makedirs(bundle, exist_ok=True)
    with open(os.path.join(bundle, "manifest.ttl"), "w") as f:
        f.write(manifest_ttl())
    with open(os.path.join(bundle, "rt-neural-generic.ttl"), "w") as f:
        f.write(plugin_ttl())
    mdir = os.path.join(bundle, os.path.dirname(DEFAULT_MODEL))
    os.makedirs(mdir, exist_ok=True)
    src = os.path.join(ROOT, "tests", "golden", "models")
    for fn in sorted(os.listdir(src)):
        if fn.endswith(".json"):
            shutil.copy(os.path.join(src, fn), os.path.join(mdir, fn))
    if binaries:
        # the plugin binary is linked HERE, into the bundle, with rpath $ORIGIN; the HIP library sits next to it
        lib = os.path.join(ROOT, "aidadsp-lv2_amd", "lib", "libaidax_hip.so")
        if not os.path.exists(lib):
            raise FileNotFoundError(f"{lib}: run `make` first")
        shutil.copy(lib, os.path.join(bundle, "libaidax_hip.so"))
        subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wextra",
                               "-I" + os.path.join(ROOT, "include"), "-shared",
                               os.path.join(ROOT, "aidadsp-lv2_amd", "lv2", "rt_neural_generic_lv2.cpp"),
                               "-o", os.path.join(bundle, "rt-neural-generic.so"),
                               "-L" + bundle, "-laidax_hip", "-Wl,-rpath,$ORIGIN"])
    return bundle


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "build"))
    ap.add_argument("--no-binaries", action="store_true")
    a = ap.parse_args()
    print(make_bundle(a.out, not a.no_binaries))
