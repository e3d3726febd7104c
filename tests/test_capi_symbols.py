"""The C-ABI library loads on a CPU-only host, exports every symbol the header
declares, its structs agree with the ctypes mirror, and it refuses to compute
without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT
from irmv_detection_amd import _build, capi

HEADER = os.path.join(ROOT, "include", "irmv_hip.h")


@pytest.fixture(scope="module")
def lib():
    _build.build()
    return capi.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(irmv_[a-z0-9_]+)\s*\(", text))
    assert len(declared) >= 28
    bound = {name for name, _, _ in capi.SYMBOLS}
    assert declared == bound, declared ^ bound
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_layouts_match_the_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "irmv_hip.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(irmv_engine_cfg), sizeof(irmv_det), sizeof(irmv_raw_dets), sizeof(irmv_kernel_stat),'
                   'offsetof(irmv_engine_cfg, camera_matrix), offsetof(irmv_engine_cfg, weights_path), offsetof(irmv_det, rvec));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(capi.EngineCfg), C.sizeof(capi.Det), C.sizeof(capi.RawDets), C.sizeof(capi.KernelStat),
           capi.EngineCfg.camera_matrix.offset, capi.EngineCfg.weights_path.offset, capi.Det.rvec.offset]
    assert got == exp


def test_defaults_are_the_reference_constants(lib):
    cfg = capi.EngineCfg()
    lib.irmv_engine_cfg_default(C.byref(cfg))
    assert (cfg.src_width, cfg.src_height, cfg.net_size) == (1280, 1024, 640)
    assert cfg.rotate180 == 1 and cfg.swap_rb == 0 and cfg.resize_mode == capi.RESIZE_STRETCH
    assert cfg.num_slots == 3 and cfg.max_det == 100 and cfg.armor_size == capi.ARMOR_SMALL
    assert abs(cfg.camera_matrix[0] - 957.669211) < 1e-12 and abs(cfg.dist_coeffs[0] + 0.405274) < 1e-12   # config/camera_info.yaml
    assert lib.irmv_version().startswith(b"irmv_hip")


def test_bad_arguments_are_rejected_before_touching_the_gpu(lib):
    cfg = capi.EngineCfg()
    lib.irmv_engine_cfg_default(C.byref(cfg))
    h = C.c_void_p()
    cfg.net_size = 100
    assert lib.irmv_engine_create(C.byref(cfg), C.byref(h)) == capi.ERR_ARG
    assert b"net_size" in lib.irmv_last_error()
    cfg.net_size = 640
    cfg.struct_size = 4
    assert lib.irmv_engine_create(C.byref(cfg), C.byref(h)) == capi.ERR_ARG
    assert lib.irmv_engine_create(None, C.byref(h)) == capi.ERR_ARG
    assert lib.irmv_engine_src_buffer(None, 0) is None or not lib.irmv_engine_src_buffer(None, 0)


def test_no_cpu_fallback(lib, blob):
    """Without a HIP device the product path must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from irmv_detection_amd.engine import PnPSolver, YoloEngine
    with pytest.raises(capi.IrmvError) as ei:
        YoloEngine(None, (1280, 1024), weights_blob=blob)
    assert ei.value.code == capi.ERR_HIP
    with pytest.raises(capi.IrmvError):
        PnPSolver([1, 0, 0, 0, 1, 0, 0, 0, 1], [0] * 5)


def test_numa_cpulist_parser_and_thread_binding(lib):
    """Host-side NUMA placement (csrc/numa.hpp): the sysfs cpulist parser, and binding a thread to a node's CPUs within the
    process's own cpuset.  Runs in a child process: affinity is per thread, and the test runner's must stay as it is."""
    assert capi.numa_parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert capi.numa_parse_cpulist("5") == [5]
    assert capi.numa_parse_cpulist("") == [] and capi.numa_parse_cpulist("\n") == []
    assert capi.numa_parse_cpulist("0-1, 4-5") == [0, 1, 4, 5]
    assert capi.numa_parse_cpulist("3-1") == []                    # malformed range: nothing is bound rather than something wrong
    assert capi.numa_parse_cpulist("0-2,x,7") == [0, 1, 2]         # parsing stops at the first thing it does not understand
    assert len(capi.numa_parse_cpulist("0-127")) == 128
    node0 = "/sys/devices/system/node/node0/cpulist"
    if not os.path.exists(node0):
        pytest.skip("no NUMA topology in sysfs")
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "from irmv_detection_amd import capi\n"
            "L = capi.load()\n"
            "before = os.sched_getaffinity(0)\n"
            "want = set(capi.numa_parse_cpulist(open(%r).read())) & before\n"
            "rc = L.irmv_numa_bind_thread(0)\n"
            "after = os.sched_getaffinity(0)\n"
            "assert (rc == 0 and after == want) or (rc != 0 and after == before and not want), (rc, before, after, want)\n"
            "assert L.irmv_numa_bind_thread(4095) != 0 and os.sched_getaffinity(0) == after\n"   # no such node: refused, nothing changed
            "buf = bytearray(8192); import ctypes as C\n"
            "n = L.irmv_numa_page_node(C.addressof(C.c_char.from_buffer(buf)))\n"
            "assert n >= -1\n"
            "print('ok', rc, sorted(after)[:4], n)\n") % (ROOT, node0)
    out = subprocess.check_output([os.sys.executable, "-c", code], text=True)
    assert out.startswith("ok")
