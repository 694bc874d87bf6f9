"""The synthetic models and signals of the parity tests: defined in the package (aidadsp-lv2_amd/workloads.py),
where bench.py finds them too; re-exported here under the name the tests use."""
import importlib

_w = importlib.import_module("aidadsp-lv2_amd.workloads")
globals().update({k: v for k, v in vars(_w).items() if not k.startswith("__")})
