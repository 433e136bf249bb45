#!/usr/bin/env python3
"""Finds, for every field the device library decodes from the bulk metrics table, the byte offsets of the raw sysfs
table that hold the same value in every (decoded, raw) pair of scripts/probes/metrics_layout_probe.cc.
usage: metrics_layout_match.py pairs.json [more.json ...]"""
import json
import sys


def candidates(snaps, values, widths=(2, 4, 8)):
    """Offsets (and widths) at which raw holds values[i] in snapshot i, for all i."""
    out = []
    for w in widths:
        if any(v >= 1 << (8 * w) for v in values):
            continue
        for off in range(4, len(snaps[0]) - w + 1, 2):
            if all(int.from_bytes(raw[off:off + w], "little") == v for raw, v in zip(snaps, values)):
                out.append((off, w))
    return out


def main():
    shots = []
    for p in sys.argv[1:]:
        shots += json.load(open(p))
    raws = [bytes.fromhex(s["raw"]) for s in shots]
    print("pairs", len(shots), "raw bytes", len(raws[0]), "decoded struct bytes", shots[0]["sizeof"])
    INVALID = {0xffff, 0xffffffff, 0xffffffffffffffff}
    for name in shots[0]:
        if name in ("raw", "sizeof"):
            continue
        v0 = shots[0][name]
        if isinstance(v0, dict):
            for sub in v0:
                sv = v0[sub]
                n = len(sv) if isinstance(sv, list) else 1
                for i in range(n):
                    vals = [(s[name][sub][i] if isinstance(sv, list) else s[name][sub]) for s in shots]
                    if all(v in INVALID for v in vals):
                        continue
                    c = candidates(raws, vals)
                    print(f"{name}.{sub}[{i}] values {vals[:3]}.. -> {c[:6]}{' ...' if len(c) > 6 else ''}")
            continue
        n = len(v0) if isinstance(v0, list) else 1
        for i in range(n):
            vals = [(s[name][i] if isinstance(v0, list) else s[name]) for s in shots]
            if all(v in INVALID for v in vals):
                print(f"{name}{'[%d]' % i if isinstance(v0, list) else ''}: not reported")
                continue
            c = candidates(raws, vals)
            print(f"{name}{'[%d]' % i if isinstance(v0, list) else ''} values {vals[:4]} -> {c[:6]}{' ...' if len(c) > 6 else ''}")


if __name__ == "__main__":
    main()
