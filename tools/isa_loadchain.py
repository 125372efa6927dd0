"""Memory-op skeleton of compiled kernels: the order of global / buffer loads (L), LDS-DMA loads (D), s_waitcnt vmcnt(n) (wN), stores
(S), barriers (|), branches (b) and labels (:) -- a quick way to see DEPENDENT load chains (L w0 L w0 ...: every load waits for the one
before it, one memory round trip each), which is how the attention kernels' resident staging spent most of its time until round 4.
   python tools/isa_loadchain.py csrc/layernorm.hip [name-substring ...] [--bf16]"""
import os
import re
import subprocess
import sys
import tempfile


def skeleton(src, subs, fp16=True):
    out = tempfile.mktemp(suffix=".s")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S", "--offload-device-only", os.path.abspath(src), "-o", out]
    if fp16:
        cmd.insert(1, "-DMH_FP16")
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=os.path.dirname(os.path.abspath(src)))
    s = open(out).read().split("\n")
    os.unlink(out)
    names = [(i, l.split(":")[0]) for i, l in enumerate(s) if re.match(r"^_Z\S+:", l)]
    for k, (i, n) in enumerate(names):
        if subs and not any(x in n for x in subs):
            continue
        end = names[k + 1][0] if k + 1 < len(names) else len(s)
        seq = []
        for l in s[i:end]:
            t = l.strip()
            if t.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
                seq.append("D" if " lds" in t else "L")
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith(("global_store", "buffer_store", "flat_store")):
                seq.append("S")
            elif t.startswith("global_atomic"):
                seq.append("A")
            elif t.startswith("s_barrier"):
                seq.append("|")
            elif t.startswith("s_cbranch") or t.startswith("s_branch"):
                seq.append("b")
            elif re.match(r"^\.LBB", t):
                seq.append(":")
            elif "s_endpgm" in t:
                seq.append("$")
        try:
            dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = n
        print(dem[:150])
        print("    " + " ".join(seq))


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skeleton(args[0], args[1:], fp16="--bf16" not in sys.argv)
