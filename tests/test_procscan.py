"""The process list without the KFD process table: the walk over /proc/<pid>/fdinfo (amdgpu's per-file DRM statistics)
and, where the kernel writes none, over the open render-node descriptors (core/backend/procscan.cc).  The reference
gets the list from the driver library and names processes from /proc/<pid>/comm
(/root/reference/bindings/go/nvml/bindings.go:491-582, 637-649); here a fake /proc tree stands in for the kernel."""
import os

import pytest

from k8s_gpu_monitor_amd import engine as E

UNKNOWN = 0xFFFFFFFFFFFFFFFF

FDINFO = """pos:\t0
flags:\t02100002
mnt_id:\t24
ino:\t135
drm-driver:\tamdgpu
drm-client-id:\t{client}
drm-pdev:\t{pdev}
pasid:\t32771
drm-memory-vram:\t{vram} KiB
drm-memory-gtt: \t{gtt} KiB
drm-memory-cpu: \t0 KiB
amd-evicted-vram:\t0 KiB
drm-engine-gfx:\t{gfx} ns
drm-engine-compute:\t{compute} ns
drm-engine-enc:\t0 ns
"""
PLAIN = "pos:\t0\nflags:\t02100002\nmnt_id:\t24\nino:\t135\n"


def _proc(root, pid, comm, fds):
    """fds: {fd number: (link target, fdinfo text)}"""
    d = root / str(pid)
    (d / "fd").mkdir(parents=True)
    (d / "fdinfo").mkdir()
    (d / "comm").write_text(comm + "\n")
    for fd, (target, info) in fds.items():
        os.symlink(target, d / "fd" / str(fd))
        (d / "fdinfo" / str(fd)).write_text(info)


def test_fdinfo_source_sums_clients_and_keeps_cards_apart(tmp_path):
    root = tmp_path / "proc"
    a, b = "0000:a7:00.0", "0000:c7:00.0"
    # pid 100: one client seen through two descriptors (dup) + a second client on the same card + one on another card
    _proc(root, 100, "trainer", {
        3: ("/dev/kfd", PLAIN),
        4: ("/dev/dri/renderD128", FDINFO.format(client=7, pdev=a, vram=1048576, gtt=2048, gfx=0, compute=5_000_000)),
        5: ("/dev/dri/renderD128", FDINFO.format(client=7, pdev=a, vram=1048576, gtt=2048, gfx=0, compute=5_000_000)),
        6: ("/dev/dri/renderD128", FDINFO.format(client=9, pdev=a, vram=512, gtt=0, gfx=100, compute=900)),
        7: ("/dev/dri/renderD129", FDINFO.format(client=11, pdev=b, vram=4096, gtt=0, gfx=0, compute=1)),
        8: ("/tmp/log", PLAIN)})
    _proc(root, 200, "other", {4: ("/dev/dri/renderD129", FDINFO.format(client=12, pdev=b, vram=64, gtt=0, gfx=0, compute=0))})
    _proc(root, 300, "shell", {0: ("/dev/pts/0", PLAIN)})
    (root / "self").mkdir()            # not a pid
    (root / "400").mkdir()             # a process whose fd table cannot be listed (no fd directory)

    src, procs = E.proc_scan(str(root), a, 128)
    assert src == "fdinfo" and [p["pid"] for p in procs] == [100]
    p = procs[0]
    assert p["name"] == "trainer" and p["vram_bytes"] == (1048576 + 512) << 10 and p["gtt_bytes"] == 2048 << 10
    assert p["gfx_engine_ns"] == 5_000_000 + 100 + 900     # the duplicate descriptor of client 7 counts once
    src, procs = E.proc_scan(str(root), b.upper(), 129)     # the address is compared without regard to case
    assert src == "fdinfo" and sorted((p["pid"], p["vram_bytes"]) for p in procs) == [(100, 4096 << 10), (200, 64 << 10)]
    # the card's address unknown: the render node alone decides
    src, procs = E.proc_scan(str(root), "", 129)
    assert sorted(p["pid"] for p in procs) == [100, 200]


def test_open_files_source_where_the_kernel_writes_no_drm_keys(tmp_path):
    """What the guests of this build's GPU pool offer: fdinfo carries pos/flags/mnt_id/ino only.  The holders of the
    card's render node are still listed, with their memory marked unknown rather than zero."""
    root = tmp_path / "proc"
    _proc(root, 398, "python", {4: ("/dev/kfd", PLAIN), 8: ("/dev/dri/renderD128", PLAIN)})
    _proc(root, 399, "bash", {0: ("/dev/null", PLAIN)})
    _proc(root, 500, "elsewhere", {8: ("/dev/dri/renderD130", PLAIN)})
    src, procs = E.proc_scan(str(root), "0000:a7:00.0", 128)
    assert src == "open-files" and len(procs) == 1
    assert procs[0]["pid"] == 398 and procs[0]["name"] == "python" and procs[0]["vram_bytes"] == UNKNOWN
    # nobody holds the card: an empty list, which is an answer
    src, procs = E.proc_scan(str(root), "0000:ff:00.0", 140)
    assert src == "open-files" and procs == []
    # no /proc at all: no source
    assert E.proc_scan(str(tmp_path / "nowhere"), "0000:a7:00.0", 128) == ("none", [])


def test_own_process_through_the_real_proc_is_harmless_without_a_gpu():
    src, procs = E.proc_scan("/proc", "0000:00:00.0", 12345)
    assert src in ("open-files", "fdinfo") and procs == []


def test_engine_names_the_source_of_its_process_list(engine2):
    assert E.process_source(0) == "fake"


def test_unknown_memory_is_none_in_the_data_model_and_na_in_ps():
    from k8s_gpu_monitor_amd.models import device as D
    assert D._mib(UNKNOWN) is None and D._mib(3 << 20) == 3
