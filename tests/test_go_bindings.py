"""The Go bindings are not compiled in this environment (no Go toolchain), so they are checked on the text against
the C headers: declared names, call arity, struct members, syntax sanity, imports, duplicate definitions and the
samples' use of the packages (k8s_gpu_monitor_amd/utils/gocheck.py).  The second half seeds one defect of each
kind into a copy of the tree and expects the checker to name it, so a clean result is not blindness.

The reference compiles and runs its Go tests against real hardware (bindings/go/nvml/nvml_test.go,
bindings/go/dcgm/dcgm_test.go); those tests are mirrored in bindings/go/*/ *_test.go and run where Go exists."""
import os
import re
import shutil
from collections import defaultdict

import pytest

from k8s_gpu_monitor_amd.utils import gocheck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "bindings", "go")
INC = os.path.join(ROOT, "include")


def test_header_parser_sees_the_whole_c_api():
    d = gocheck.parse_c(open(os.path.join(INC, "gpumon.h")).read())
    # every exported prototype of the header, found independently with a line regex
    text = gocheck.strip_c_comments(open(os.path.join(INC, "gpumon.h")).read())
    protos = set(re.findall(r"^GM_API\s+[\w\s\*]+?\b(gm_\w+)\s*\(", text, flags=re.M))
    assert len(protos) > 80 and protos <= set(d.functions), sorted(protos - set(d.functions))
    assert d.functions["gm_init"] == 1 and d.functions["gm_shutdown"] == 0 and d.functions["gm_xgmi_links"] == 3
    assert d.functions["gm_policy_register"] == 5 and d.functions["gm_values_since"] == 5
    assert {"GM_ST_OK", "GM_HEALTH_WATCH_XGMI", "GM_FI_XGMI_ERROR_STATUS", "GM_MAX_XGMI_LINKS", "GM_EVENT_DIAG_FAIL"} <= d.constants
    assert {"gm_status_t", "gm_group_t", "gm_violation_cb", "gm_xgmi_link_t", "gm_value_t"} <= d.types
    assert set(d.structs["gm_xgmi_link_t"]) == {"link", "state", "peer_gpu", "peer_bdf", "bit_rate_gbps", "max_bandwidth_gbps",
                                               "rx_kb", "tx_kb"}
    assert d.structs["gm_topology_t"]["peers"] == "gm_peer_link_t"          # nested member types resolve
    assert "v" in d.structs["gm_value_t"] and "d" in d.structs["gm_violation_t"]  # unions count as one member


def test_go_tree_is_consistent_with_the_headers():
    stats = defaultdict(int)
    problems = gocheck.check_tree(GO, [INC], stats)
    assert problems == [], "\n".join(problems)
    # the check looked at the tree: counts well below these mean a parser that went blind
    assert stats["go_files"] >= 30 and stats["c_references"] >= 350
    assert stats["calls_with_arity_checked"] >= 100 and stats["member_accesses_checked"] >= 200
    assert stats["package_references_checked"] >= 70


def test_every_c_function_of_the_new_apis_is_bound():
    """Links, health thresholds and diagnostics verdicts are reachable from Go."""
    used = set()
    for dirpath, _d, files in os.walk(GO):
        for f in files:
            if f.endswith(".go"):
                used |= set(re.findall(r"C\.(gm_\w+)", gocheck.strip_go(open(os.path.join(dirpath, f)).read())))
    assert {"gm_xgmi_links", "gm_smi_xgmi_links", "gm_health_thresholds_get", "gm_health_thresholds_set",
            "gm_diag_result_submit", "gm_diag_results"} <= used


def _mutate(tmp_path, rel, old, new, count=1):
    root = tmp_path / "go"
    if not root.exists():
        shutil.copytree(GO, root)
        shutil.copytree(INC, tmp_path / "include")
    p = root / rel if not rel.startswith("include/") else tmp_path / rel
    s = p.read_text()
    assert old in s, (rel, old)
    p.write_text(s.replace(old, new, count))
    return gocheck.check_tree(str(root), [str(tmp_path / "include")])


SEEDED = [
    # (file, old, new, fragment of the expected message)
    ("hostengine/links.go", "C.gm_xgmi_links(", "C.gm_xgmi_linkz(", "C.gm_xgmi_linkz is not declared"),
    ("hostengine/links.go", "C.gm_xgmi_links(C.uint32_t(gpuID), &raw[0], &n)", "C.gm_xgmi_links(C.uint32_t(gpuID), &raw[0])",
     "C.gm_xgmi_links takes 3 argument(s), called with 2"),
    ("hostengine/links.go", "l.peer_bdf[0]", "l.peer_bus[0]", "gm_xgmi_link_t has no member 'peer_bus'"),
    ("hostengine/topology.go", "p.max_bw_mbs >= 0", "p.max_bandwidth >= 0", "gm_peer_link_t has no member 'max_bandwidth'"),
    ("hostengine/health.go", "t.mem_fail_c", "t.hbm_fail_c", "gm_health_thresholds_t has no member 'hbm_fail_c'"),
    ("hostengine/health.go", "C.GM_HEALTH_WATCH_XGMI", "C.GM_HEALTH_WATCH_NVLINK", "C.GM_HEALTH_WATCH_NVLINK is not declared"),
    ("hostengine/diag.go", '\t"math"\n', "", 'package "math" is used but not imported'),
    ("hostengine/diag.go", '\t"math"\n', '\t"math"\n\t"sort"\n', 'import "sort" is not used'),
    ("hostengine/diag.go", "\treturn out, nil\n}", "\treturn out, nil\n", "unclosed '{'"),
    ("hostengine/diag.go", "func GetDiagResults(", "func GetXgmiLinks(", "GetXgmiLinks is defined in"),
    ("smi/bindings.go", "C.smi_value_i64(", "C.smi_value_int(", "C.smi_value_int is not declared"),
    ("smi/bindings.go", "C.gm_smi_peer_link(C.uint32_t(a), C.uint32_t(b), &p)", "C.gm_smi_peer_link(C.uint32_t(a), &p)",
     "C.gm_smi_peer_link takes 3 argument(s), called with 2"),
    ("samples/hostengine/topology/main.go", "he.GetSupportedDevices()", "he.GetSupportedGPUs()", "he.GetSupportedGPUs is not exported"),
    ("hostengine/policy.go", "//export goViolationNotify", "//export goViolationNotice", "goViolationNotice without a Go function"),
    ("hostengine/capi.h", "static inline void he_thresholds(", "static inline void he_thresholdz(", "C.he_thresholds is not declared"),
    ("include/gpumon.h", "GM_API gm_status_t gm_diag_results(uint32_t gpu, gm_diag_result_t* out, int* n);", "",
     "C.gm_diag_results is not declared"),
    ("include/gpumon.h", "int64_t rx_kb, tx_kb;      /* cumulative; -1 = blank */", "int64_t rx_kb;", "gm_xgmi_link_t has no member 'tx_kb'"),
    ("include/gpumon.h", "GM_API gm_status_t gm_health_set(gm_group_t g, uint32_t systems);",
     "GM_API gm_status_t gm_health_set(gm_group_t g, uint32_t systems, int flags);", "C.gm_health_set takes 3 argument(s), called with 2"),
]


@pytest.mark.parametrize("rel,old,new,expect", SEEDED, ids=[f"{s[0]}:{s[3][:40]}" for s in SEEDED])
def test_checker_names_a_seeded_defect(tmp_path, rel, old, new, expect):
    problems = _mutate(tmp_path, rel, old, new)
    assert any(expect in p for p in problems), "\n".join(problems) or "no problem reported"
    # and nothing unrelated is reported alongside (the file that was changed, or its dependants)
    assert len(problems) <= 6, "\n".join(problems)


def test_strip_go_keeps_positions_and_hides_literals():
    src = 'package x\n// C.bogus(\nvar s = "C.fake(1,2" + `C.raw(\n)` + \'"\'\nfunc f() { C.real(1) }\n'
    out = gocheck.strip_go(src)
    assert len(out) == len(src) and out.count("\n") == src.count("\n")
    assert "bogus" not in out and "fake" not in out and "raw" not in out and "C.real(1)" in out
