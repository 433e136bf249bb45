"""Test oracle: ground truth from the ``amd-smi`` command-line tool.

The reference checks its bindings against ``nvidia-smi --query-gpu=... --format=csv``
(bindings/go/nvml/nvsmi/nvsmi.go:17-43).  Here the independent oracle is
``amd-smi <static|metric> --json``; the helpers flatten its nested
``{"value": v, "unit": u}`` leaves and "N/A" strings into plain numbers / None so a
test can compare field by field with tolerances.
"""
import json
import shutil
import subprocess

_BIN_CANDIDATES = ("amd-smi", "/opt/rocm/bin/amd-smi")


def binary():
    for b in _BIN_CANDIDATES:
        p = shutil.which(b)
        if p:
            return p
    return None


def available():
    return binary() is not None


def _leaf(v):
    """{"value": x, "unit": u} -> x ; "N/A" -> None ; numeric strings -> numbers."""
    if isinstance(v, dict) and "value" in v and set(v) <= {"value", "unit"}:
        return _leaf(v["value"])
    if isinstance(v, str):
        s = v.strip()
        if s.upper() in ("N/A", "NA", ""):
            return None
        try:
            return int(s, 0)
        except ValueError:
            pass
        try:
            return float(s)
        except ValueError:
            return s
    return v


def flatten(obj, prefix=""):
    """Nested JSON -> {"a.b.c": leaf}."""
    out = {}
    if isinstance(obj, dict) and not (set(obj) <= {"value", "unit"} and "value" in obj):
        for k, v in obj.items():
            out.update(flatten(v, f"{prefix}{k}."))
    elif isinstance(obj, list):
        for i, v in enumerate(obj):
            out.update(flatten(v, f"{prefix}{i}."))
    else:
        out[prefix[:-1]] = _leaf(obj)
    return out


def parse(text):
    """amd-smi prints either a list of per-GPU objects or {"gpu_data": [...]}: returns that list."""
    start = min((i for i in (text.find("{"), text.find("[")) if i >= 0), default=-1)
    if start < 0:
        raise ValueError("no JSON in amd-smi output")
    doc = json.loads(text[start:text.rfind("}" if text.rfind("}") > text.rfind("]") else "]") + 1])
    if isinstance(doc, dict):
        doc = doc.get("gpu_data", [doc])
    return doc


def query(section, gpu=None, extra=(), timeout=60):
    """Runs ``amd-smi <section> [-g gpu] --json`` and returns [{flattened fields}] per GPU."""
    b = binary()
    if not b:
        raise FileNotFoundError("amd-smi not found")
    cmd = [b, section] + (["-g", str(gpu)] if gpu is not None else []) + list(extra) + ["--json"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} -> {r.returncode}: {r.stderr[-400:]}")
    return [flatten(g) for g in parse(r.stdout)]


def device_count():
    return len(query("list"))


def first(d, *suffixes):
    """Value of the first key ending with one of the suffixes (amd-smi renames keys between releases)."""
    for suf in suffixes:
        for k, v in d.items():
            if k == suf or k.endswith("." + suf):
                return v
    return None
