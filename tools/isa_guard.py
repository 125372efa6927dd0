#!/usr/bin/env python3
"""Build-time performance guard for the hot kernels.

Two accidents of round 3 cost 12 % on every GEMM for most of the round and were found by chance: a never-taken branch
that changed the default kernel's register allocation (112 -> 90 VGPRs), and a waterfall loop (v_readfirstlane x 4 +
s_and_saveexec) around LDS-DMA loads inside the K loop.  This tool reads the gfx950 code objects out of the built
objects (csrc/*.o: `.hip_fatbin` section -> clang-offload-bundler) and checks, for every kernel named in
tools/isa_bands.json:

  * `.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size`,
    `.group_segment_fixed_size` from the code-object notes against the committed band (min / max per field);
  * `load_chain`: the longest run of dependent memory round trips (load - s_waitcnt vmcnt(0) - load - ...; round 4's second session found the
    attention staging, LayerNorm rows and a dozen small kernels spending most of their time in such chains: predicated loads the compiler
    serialises) must not grow beyond the committed number;
  * the disassembly: the kernel's hot loop -- the innermost backward-branch region that contains MFMA instructions -- must hold
    its committed MFMA count and no `scratch_` access (a spill inside the loop); and the number of WATERFALL loops anywhere in the
    kernel (a short `s_cbranch_execnz` loop around `v_readfirstlane` + `s_and_saveexec` + a memory instruction: hipcc's wrapper for a
    buffer descriptor it cannot prove wave-uniform, cdna_hip_programming.md T20) must not exceed the committed number.

`python tools/isa_guard.py` checks (exit code 1 + a report on any violation); `--update` rewrites the bands from the current
build (review the diff before committing it); `--show NAME` prints one kernel's numbers and loop.  CPU-only: runs in
__graft_entry__.build().
"""
from __future__ import annotations

import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multimodal_propaganda_meme_classification_amd", "csrc")
BANDS = os.path.join(ROOT, "tools", "isa_bands.json")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size")
FORBIDDEN_IN_LOOP = ("scratch_load", "scratch_store")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def code_object(obj: str, tmp: str) -> str:
    fat = os.path.join(tmp, os.path.basename(obj) + ".fat")
    co = os.path.join(tmp, os.path.basename(obj) + ".co")
    _run(f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", obj)
    _run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}")
    return co


def demangle_names(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, out))


def notes(co: str):
    """kernel symbol -> {field: int} from the AMDGPU metadata note (amdhsa.kernels: one `  - .key: value` list entry per kernel)"""
    txt = _run(f"{LLVM}/llvm-readelf", "--notes", co)
    entries, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"^  - \.(\w+):\s*(.*)$", line)
        if m:
            cur = {m.group(1): m.group(2).strip()}
            entries.append(cur)
            continue
        m = re.match(r"^    \.(\w+):\s*(.*)$", line)
        if m and cur is not None:
            cur[m.group(1)] = m.group(2).strip()
    res = {}
    for k in entries:
        if k.get("symbol", "").endswith(".kd"):
            res[k["symbol"][:-3]] = {f: int(k.get(f, "0")) for f in FIELDS}
    return res


def disassemble(co: str):
    """kernel symbol -> list of (addr, text) instructions"""
    txt = _run(f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co)
    funcs, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            cur = funcs.setdefault(m.group(2), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s*// ([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur.append((int(m.group(2), 16), m.group(1)))
    return funcs


def hot_loop(insns):
    """innermost loop (backward branch target..branch) that holds the most MFMA instructions per byte span"""
    addr_index = {a: i for i, (a, _) in enumerate(insns)}
    best = None
    for i, (a, t) in enumerate(insns):
        m = re.match(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", t)
        if not m:
            continue
        # llvm-objdump prints the branch as a signed 16-bit dword offset
        off = int(m.group(1) or m.group(2))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt > a or tgt not in addr_index:
            continue
        j = addr_index[tgt]
        body = insns[j:i + 1]
        n_mfma = sum(1 for _, x in body if x.startswith("v_mfma"))
        if n_mfma == 0:
            continue
        # innermost = fewest instructions among loops with MFMAs; prefer the one with the most MFMAs when nested loops tie
        key = (len(body), -n_mfma)
        if best is None or key < best[0]:
            best = (key, body)
    return best[1] if best else []


def waterfall_loops(insns):
    """count of hipcc waterfall loops: a backward s_cbranch_execnz over <= 40 instructions holding v_readfirstlane, s_and_saveexec and a VMEM op"""
    addr_index = {a: i for i, (a, _) in enumerate(insns)}
    n = 0
    for i, (a, t) in enumerate(insns):
        m = re.match(r"s_cbranch_execnz\s+(\d+)", t)
        if not m:
            continue
        off = int(m.group(1))
        if off >= 32768:
            off -= 65536
        tgt = a + 4 + off * 4
        if tgt > a or tgt not in addr_index:
            continue
        body = [x for _, x in insns[addr_index[tgt]:i + 1]]
        if len(body) <= 40 and any(x.startswith("v_readfirstlane") for x in body) and any(x.startswith("s_and_saveexec") for x in body) \
                and any(x.startswith(("buffer_", "global_", "flat_")) for x in body):
            n += 1
    return n


def load_chain(insns):
    """longest run of DEPENDENT memory round trips in program order: a VMEM load followed by `s_waitcnt vmcnt(0)` before the next VMEM
    load, repeated (the L w0 L w0 ... pattern of tools/isa_loadchain.py: predicated loads the compiler serialised, one-load-per-trip
    loops).  Branches and labels are skipped; a store, a barrier or a counted wait ends a run."""
    best = cur = 0
    pending = False
    for _, t in insns:
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            pending = True
            continue
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m and int(m.group(1)) == 0 and pending:
                cur += 1
                best = max(best, cur)
            elif m and int(m.group(1)) != 0:
                cur = 0
            pending = False
            continue
        if t.startswith(("global_store", "buffer_store", "flat_store", "s_barrier", "global_atomic")):
            cur = 0
            pending = False
    return best


def collect():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for fn in sorted(os.listdir(CSRC)):
            if not fn.endswith(".o"):
                continue
            co = code_object(os.path.join(CSRC, fn), tmp)
            nt = notes(co)
            dis = disassemble(co)
            dm = demangle_names(list(nt))
            for sym, fields in nt.items():
                name = dm[sym]
                name = re.sub(r"\(anonymous namespace\)::", "", name)
                name = re.sub(r"^void ", "", name)
                name = re.sub(r"\(.*\)$", "", name)
                loop = hot_loop(dis.get(sym, []))
                rec = dict(fields)
                rec["loop_insns"] = len(loop)
                rec["loop_mfma"] = sum(1 for _, x in loop if x.startswith("v_mfma"))
                rec["loop_forbidden"] = sorted({f for _, x in loop for f in FORBIDDEN_IN_LOOP if x.startswith(f)})
                rec["waterfall_loops"] = waterfall_loops(dis.get(sym, []))
                rec["load_chain"] = load_chain(dis.get(sym, []))
                rec["_loop"] = loop
                out[f"{fn}:{name}"] = rec
    return out


def main(argv):
    got = collect()
    if "--show" in argv:
        pat = argv[argv.index("--show") + 1]
        for k, r in got.items():
            if pat in k:
                print(k, {f: r[f] for f in FIELDS}, "loop:", r["loop_insns"], "insns,", r["loop_mfma"], "mfma", r["loop_forbidden"], "waterfall loops:", r["waterfall_loops"],
                      "load chain:", r["load_chain"])
                if "--loop" in argv:
                    for a, t in r["_loop"]:
                        print(f"    {a:08x}  {t}")
        return 0
    if "--update" in argv:
        old = json.load(open(BANDS)) if os.path.exists(BANDS) else {"kernels": {}}
        names = list(old["kernels"]) if old["kernels"] and "--all" not in argv else []
        for a in argv:
            if a.startswith("--add="):
                names += [k for k in got if a[6:] in k]
        bands = {"comment": "committed resource bands of the hot kernels; tools/isa_guard.py fails the build when a kernel leaves them",
                 "kernels": {}}
        for k in sorted(set(names)):
            if k not in got:
                print("dropped (no longer built):", k)
                continue
            r = got[k]
            v = r["vgpr_count"]
            vhi = v + 4
            for edge in (64, 96, 128, 168, 256):      # never across an occupancy step of the unified register file (8 / 5 / 4 / 3 / 2 waves per SIMD)
                if v <= edge < vhi:
                    vhi = edge
            bands["kernels"][k] = {"vgpr_count": [max(0, v - 4), vhi], "agpr_count": [0, r["agpr_count"]],
                                   "sgpr_count": [0, max(r["sgpr_count"], 96)],
                                   "vgpr_spill_count": [0, 0], "sgpr_spill_count": [0, r["sgpr_spill_count"]],
                                   "private_segment_fixed_size": [0, r["private_segment_fixed_size"]],
                                   "group_segment_fixed_size": [0, r["group_segment_fixed_size"]],
                                   "loop_mfma": [r["loop_mfma"], r["loop_mfma"]], "waterfall_loops": [0, r["waterfall_loops"]],
                                   "load_chain": [0, r["load_chain"]],
                                   "loop_forbidden": []}
        json.dump(bands, open(BANDS, "w"), indent=1, sort_keys=True)
        print(f"wrote {len(bands['kernels'])} kernels to {BANDS}")
        return 0
    bands = json.load(open(BANDS))["kernels"]
    bad = []
    for k, b in bands.items():
        if k not in got:
            bad.append(f"{k}: not found in the build")
            continue
        r = got[k]
        for f, (lo, hi) in ((f, b[f]) for f in b if f not in ("loop_forbidden",)):
            if not (lo <= r[f] <= hi):
                bad.append(f"{k}: {f} = {r[f]} outside [{lo}, {hi}]")
        extra = [x for x in r["loop_forbidden"] if x not in b.get("loop_forbidden", [])]
        if extra:
            bad.append(f"{k}: hot loop contains {extra} (waterfall loop / spill)")
    if bad:
        print("isa_guard: FAILED")
        for x in bad:
            print("  ", x)
        return 1
    print(f"isa_guard: {len(bands)} kernels inside their bands")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
