"""CPU checks of the drop-in boundary: the C-ABI library builds, loads, and exports exactly the
symbols include/memehip.h declares; the Python binding covers all of them; and the product refuses
to run without a HIP device (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "memehip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mh_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from multimodal_propaganda_meme_classification_amd import _lib
    return _lib


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 20
    for path in (lib.LIB_PATH, lib.LIB_PATH_F16):           # the bf16 and the fp16 build export the same ABI
        cdll = ctypes.CDLL(path)
        for n in names:
            assert hasattr(cdll, n), f"{n} declared in include/memehip.h but not exported by {path}"


def test_binding_covers_header(lib):
    assert sorted(lib.EXPORTED_SYMBOLS) == _declared()
    handle = lib.load()
    assert handle.mh_version().startswith(b"memehip") and b"bf16" in handle.mh_version()
    assert b"fp16" in lib.load("fp16").mh_version()
    assert b"shape" in handle.mh_status_str(2)


def test_struct_layout_matches_header(lib):
    # 8 pointers + 7 int32 + float + pointer + float + uint32 + 2 pointers = 128 bytes; a drift here would corrupt every grouped launch
    assert ctypes.sizeof(lib.MhGemmProblem) == 136
    assert ctypes.sizeof(lib.MhColsumJob) == 24
    assert ctypes.sizeof(lib.MhAttnProblem) == 104
    assert ctypes.sizeof(lib.MhLnFwdJob) == 72 and ctypes.sizeof(lib.MhLnBwdJob) == 112
    assert ctypes.sizeof(lib.MhHeadParams) == 64 == ctypes.sizeof(lib.MhHeadGrads)


def test_no_cpu_fallback(lib):
    from multimodal_propaganda_meme_classification_amd import ops
    x = torch.zeros((128, 64), dtype=torch.bfloat16)
    w = torch.zeros((128, 64), dtype=torch.bfloat16)
    with pytest.raises(lib.MemehipError):
        ops.linear_fwd(x, w)
    with pytest.raises(lib.MemehipError):
        ops.layernorm_fwd(x, torch.ones(64), torch.zeros(64), 1e-6)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "multimodal_propaganda_meme_classification_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "from .. import oracle" not in src and "import oracle" not in src, f


def test_ctypes_structs_match_the_c_compiler(tmp_path):
    """Every struct of include/memehip.h, sized by gcc, against its ctypes mirror: a field added on one side only shifts the
    second element of a job array and fails far from its cause."""
    import ctypes
    import subprocess
    from multimodal_propaganda_meme_classification_amd import _lib as lib
    names = ["MhGemmProblem", "MhColsumJob", "MhLnFwdJob", "MhLnBwdJob", "MhAttnProblem", "MhHeadParams", "MhHeadGrads", "MhGemmF32", "MhConvPackJob", "MhConvWgradJob", "MhConvGeom", "MhConvWgradProblem", "MhConvBnBwd", "MhAdamSkipGroups", "MhLossScale"]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "memehip.h"\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n in names:
        assert ctypes.sizeof(getattr(lib, n)) == int(sizes[n]), (n, ctypes.sizeof(getattr(lib, n)), sizes[n])


def test_isa_guard_bands_hold(lib):
    """tools/isa_guard.py (run by __graft_entry__.build()): register counts, spills, scratch and the K-loop ISA of the hot kernels
    are inside the committed bands -- and the guard does fire: a band moved away from the build must fail."""
    import json
    import subprocess
    import sys
    guard = os.path.join(ROOT, "tools", "isa_guard.py")
    r = subprocess.run([sys.executable, guard], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    bands = json.load(open(os.path.join(ROOT, "tools", "isa_bands.json")))["kernels"]
    assert any("gemm_kernel<0, 0, false>" in k for k in bands) and any("conv_gemm_kernel" in k for k in bands) and any("attn_" in k for k in bands)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_guard
    got = isa_guard.collect()
    k = "gemm.f16.o:gemm_kernel<0, 0, false>"
    assert bands[k]["vgpr_count"][0] <= got[k]["vgpr_count"] <= bands[k]["vgpr_count"][1]
    assert got[k]["loop_mfma"] == 16 and got[k]["vgpr_spill_count"] == 0 and got[k]["waterfall_loops"] == 0
    # the accident of round 3 (112 -> 90 VGPRs) would have left the band
    assert not (bands[k]["vgpr_count"][0] <= 90 <= bands[k]["vgpr_count"][1])


def test_load_chain_guard_and_skeleton_tool(lib):
    """round 4, second session: the guard's `load_chain` (longest run of load - s_waitcnt vmcnt(0) - load: dependent memory round trips)
    is zero for the kernels that were rewritten for it, and tools/isa_loadchain.py prints a kernel's memory skeleton from the source."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_guard
    bands = json.load(open(os.path.join(ROOT, "tools", "isa_bands.json")))["kernels"]
    got = isa_guard.collect()
    for k in ("layernorm.f16.o:ln_fwd_kernel<2, 2>", "head.f16.o:linear_fwd_kernel", "layernorm.f16.o:colsum_partials_kernel"):
        assert bands[k]["load_chain"] == [0, 0] and got[k]["load_chain"] == 0, k
    # a synthetic chain is counted: load, wait, load, wait
    insns = [(0, "global_load_dwordx4 v[0:3], v[4:5], off"), (4, "s_waitcnt vmcnt(0)"), (8, "s_cbranch_execz 12"),
             (12, "global_load_dwordx4 v[0:3], v[4:5], off"), (16, "s_waitcnt vmcnt(0) lgkmcnt(1)"), (20, "global_store_dword v0, v1, off")]
    assert isa_guard.load_chain(insns) == 2
    assert isa_guard.load_chain([(0, "global_load_dword v0, v1, off"), (4, "global_load_dword v2, v1, off"), (8, "s_waitcnt vmcnt(1)")]) == 0
    tool = os.path.join(ROOT, "tools", "isa_loadchain.py")
    src = os.path.join(ROOT, "multimodal_propaganda_meme_classification_amd", "csrc", "dwconv.hip")
    r = subprocess.run([sys.executable, tool, src, "dwconv_weight_pack"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(lines) == 2 and "dwconv_weight_pack_kernel" in lines[0] and set(lines[1].split()) >= {"L", "S", "$"}
