#!/usr/bin/env python3
"""Extracts the port table of the reference plugin from its Turtle description into tests/golden/ports.json
(a data fixture: index, symbol, name, direction/type, default, minimum, maximum, unit, port properties,
designation, scale points of each of the 25 ports, plus the default-state model path).

Run in the build container only (the reference does not travel to the GPU box):

    python tests/golden/make_ports_json.py [/root/reference]

Source: rt-neural-generic/ttl/rt-neural-generic.ttl:61-317 of the reference."""
import json
import os
import re
import sys



def one(pattern, text, conv=str):
    m = re.search(pattern, text)
    return conv(m.group(1)) if m else None


def parse_ttl(ttl: str) -> dict:
    """Port table, features and default state of a plugin description (also used by tests/test_bundle.py on the
    TTL that tools/make_bundle.py generates, so both sides go through the same reader)."""
    body = ttl[ttl.index("lv2:port"):ttl.index("state:state")]
    # top-level [...] blocks; scale points nest one level of brackets inside a port
    blocks, depth, start = [], 0, None
    for i, ch in enumerate(body):
        if ch == "[":
            if depth == 0:
                start = i + 1
            depth += 1
        elif ch == "]":
            depth -= 1
            if depth == 0:
                blocks.append(body[start:i])
    ports = []
    for b in blocks:
        flat = re.sub(r"\[[^\]]*\]", "", b) + ";"             # properties of the port itself
        types = re.search(r"\ba\s+([^;]+);", flat).group(1)
        ports.append({
            "index": one(r"lv2:index\s+(\d+)", flat, int),
            "symbol": one(r'lv2:symbol\s+"([^"]+)"', flat),
            "name": one(r'lv2:name\s+"([^"]+)"', flat),
            "types": sorted(t.strip() for t in types.split(",")),
            "default": one(r"lv2:default\s+([-0-9.eE]+)", flat, float),
            "minimum": one(r"lv2:minimum\s+([-0-9.eE]+)", flat, float),
            "maximum": one(r"lv2:maximum\s+([-0-9.eE]+)", flat, float),
            "unit": one(r"units:unit\s+units:(\w+)", flat),
            "properties": sorted(re.findall(r"lv2:portProperty\s+lv2:(\w+)", flat)),
            "designation": one(r"lv2:designation\s+lv2:(\w+)", flat),
            "scale_points": [[lab, float(val)] for lab, val in
                             re.findall(r'lv2:scalePoint\s*\[\s*rdfs:label\s+"([^"]+)"\s*;\s*rdf:value\s+([-0-9.eE]+)\s*\]', b)],
        })
    ports.sort(key=lambda p: p["index"])
    state = re.search(r"state:state\s*\[\s*<([^>]+)>\s*<([^>]+)>", ttl)
    return {
        "plugin_uri": re.search(r"\n<([^>#]+)>\s*\n\s*a lv2:Plugin", ttl).group(1),
        "required_features": sorted(re.findall(r"(\w+:\w+)", re.search(r"lv2:requiredFeature([^;]+);", ttl).group(1))),
        "extension_data": sorted(re.findall(r"(\w+:\w+)", re.search(r"lv2:extensionData([^;]+);", ttl).group(1))),
        "default_state": {"key": state.group(1), "path": state.group(2)},
        "ports": ports,
    }


if __name__ == "__main__":
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    src = "rt-neural-generic/ttl/rt-neural-generic.ttl"
    out = {"source": src}
    out.update(parse_ttl(open(os.path.join(ref, src)).read()))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ports.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(dst, len(out["ports"]), "ports")
