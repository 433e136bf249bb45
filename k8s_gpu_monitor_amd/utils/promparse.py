"""Strict parser for the Prometheus text exposition format (version 0.0.4).

Used by the tests to validate what the exporter renders: it rejects anything a
real Prometheus server would reject or silently mis-handle — samples of one
family that are not contiguous, a sample before its TYPE line, duplicate
HELP/TYPE lines, malformed label sets, duplicate series, non-numeric values.
"""
import math
import re
from collections import OrderedDict

_NAME = re.compile(r"^[a-zA-Z_:][a-zA-Z0-9_:]*$")
_LABEL = re.compile(r"^[a-zA-Z_][a-zA-Z0-9_]*$")
_TYPES = {"counter", "gauge", "histogram", "summary", "untyped"}


class ExpositionError(ValueError):
    pass


class Family:
    def __init__(self, name):
        self.name = name
        self.help = None
        self.type = None
        self.samples = []  # (labels OrderedDict, value)


def _parse_labels(s, lineno):
    """s starts after '{'; returns (OrderedDict, rest after '}')."""
    labels = OrderedDict()
    i = 0
    while True:
        while i < len(s) and s[i] in " \t":
            i += 1
        if i < len(s) and s[i] == "}":
            return labels, s[i + 1:]
        j = s.find("=", i)
        if j < 0:
            raise ExpositionError(f"line {lineno}: label without '='")
        name = s[i:j].strip()
        if not _LABEL.match(name):
            raise ExpositionError(f"line {lineno}: bad label name {name!r}")
        if name in labels:
            raise ExpositionError(f"line {lineno}: duplicate label {name!r}")
        if j + 1 >= len(s) or s[j + 1] != '"':
            raise ExpositionError(f"line {lineno}: label value must be quoted")
        i = j + 2
        val = []
        while True:
            if i >= len(s):
                raise ExpositionError(f"line {lineno}: unterminated label value")
            c = s[i]
            if c == "\\":
                if i + 1 >= len(s) or s[i + 1] not in '\\"n':
                    raise ExpositionError(f"line {lineno}: bad escape in label value")
                val.append({"\\": "\\", '"': '"', "n": "\n"}[s[i + 1]])
                i += 2
            elif c == '"':
                i += 1
                break
            else:
                val.append(c)
                i += 1
        labels[name] = "".join(val)
        while i < len(s) and s[i] in " \t":
            i += 1
        if i < len(s) and s[i] == ",":
            i += 1
        elif i < len(s) and s[i] == "}":
            return labels, s[i + 1:]
        else:
            raise ExpositionError(f"line {lineno}: expected ',' or '}}' after label value")


def _parse_value(tok, lineno):
    low = tok.lower()
    if low in ("nan",):
        return math.nan
    if low in ("+inf", "inf"):
        return math.inf
    if low == "-inf":
        return -math.inf
    try:
        return float(tok)
    except ValueError:
        raise ExpositionError(f"line {lineno}: bad sample value {tok!r}")


def parse(text, require_type=True):
    """Returns an OrderedDict name -> Family; raises ExpositionError on any violation."""
    if text and not text.endswith("\n"):
        raise ExpositionError("exposition must end with a newline")
    families = OrderedDict()
    closed = set()  # families whose block has ended
    current = None
    seen_series = set()
    for lineno, line in enumerate(text.split("\n")[:-1], 1):
        if line == "":
            continue
        if line.startswith("#"):
            parts = line.split(None, 3)
            if len(parts) >= 3 and parts[1] in ("HELP", "TYPE"):
                name = parts[2]
                if not _NAME.match(name):
                    raise ExpositionError(f"line {lineno}: bad metric name {name!r}")
                if name in closed:
                    raise ExpositionError(f"line {lineno}: family {name} re-opened (samples must be contiguous)")
                if current is not None and current != name:
                    closed.add(current)
                fam = families.setdefault(name, Family(name))
                current = name
                if parts[1] == "HELP":
                    if fam.help is not None:
                        raise ExpositionError(f"line {lineno}: second HELP for {name}")
                    fam.help = parts[3] if len(parts) > 3 else ""
                else:
                    if fam.type is not None:
                        raise ExpositionError(f"line {lineno}: second TYPE for {name}")
                    if fam.samples:
                        raise ExpositionError(f"line {lineno}: TYPE for {name} after its samples")
                    t = parts[3].strip() if len(parts) > 3 else ""
                    if t not in _TYPES:
                        raise ExpositionError(f"line {lineno}: unknown type {t!r}")
                    fam.type = t
            continue  # other comments are ignored
        m = re.match(r"^([a-zA-Z_:][a-zA-Z0-9_:]*)(.*)$", line)
        if not m:
            raise ExpositionError(f"line {lineno}: bad sample line {line!r}")
        name, rest = m.group(1), m.group(2)
        labels = OrderedDict()
        if rest.startswith("{"):
            labels, rest = _parse_labels(rest[1:], lineno)
        toks = rest.split()
        if len(toks) not in (1, 2):
            raise ExpositionError(f"line {lineno}: expected value [timestamp]")
        value = _parse_value(toks[0], lineno)
        if len(toks) == 2 and not re.match(r"^-?\d+$", toks[1]):
            raise ExpositionError(f"line {lineno}: bad timestamp")
        if name in closed:
            raise ExpositionError(f"line {lineno}: sample of {name} outside its family block")
        if current is not None and current != name:
            closed.add(current)
        fam = families.setdefault(name, Family(name))
        current = name
        if require_type and fam.type is None:
            raise ExpositionError(f"line {lineno}: sample of {name} before its TYPE line")
        key = (name, tuple(sorted(labels.items())))
        if key in seen_series:
            raise ExpositionError(f"line {lineno}: duplicate series {name}{dict(labels)}")
        seen_series.add(key)
        fam.samples.append((labels, value))
    return families
