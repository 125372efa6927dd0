#!/usr/bin/env python3
"""Put numbers behind the parity tolerances.  GPU box, one run:

    python tools/publish_parity.py [profiles/r04_parity.txt]

runs the parity tests that compare the HIP path with the reference runs / the oracle (reference-run fixtures, config 3 at batch 32
in both dtypes, config 5, the ResNet tower, the fp16 model tests) with MEMEHIP_PARITY_OUT set, so that every measurement they make
(tests/conftest.py: parity_log -- max |hip - reference|, error / spread of the reference values, worst gradient deviation, gradient
norms per step, TSV flips, parameter movement) lands in one file, headed by the commit and the device.  The asserts in tests/ are
held to <= 2x the values recorded there (each assert's comment quotes its measured value).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TESTS = ["tests/test_reference_run_gpu.py", "tests/test_round2_gpu.py", "tests/test_config5_gpu.py", "tests/test_resnet_gpu.py",
         "tests/test_model_fp16_gpu.py"]


def main():
    out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity.txt"))
    os.makedirs(os.path.dirname(out), exist_ok=True)
    import torch
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip() or "(snapshot)"
    with open(out, "w", encoding="utf-8") as f:
        f.write(f"# parity measurements of {', '.join(TESTS)}\n# commit {head}; {torch.cuda.get_device_name(0) if torch.cuda.is_available() else 'no GPU'}; "
                f"torch {torch.__version__}\n# one line per measurement, in test order; the asserts of tests/ are held to <= 2x these values\n")
    env = dict(os.environ, MEMEHIP_PARITY_OUT=out)
    rc = 0
    for t in TESTS:
        with open(out, "a", encoding="utf-8") as f:
            f.write(f"\n## {t}\n")
        r = subprocess.run([sys.executable, "-m", "pytest", t, "-m", "gpu", "-q", "-s", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                           capture_output=True, text=True)
        tail = [ln for ln in r.stdout.splitlines() if " passed" in ln or " failed" in ln or " error" in ln]
        with open(out, "a", encoding="utf-8") as f:
            f.write(f"# pytest: {tail[-1] if tail else 'rc ' + str(r.returncode)}\n")
        if r.returncode != 0:
            rc = r.returncode
            sys.stderr.write(r.stdout[-3000:] + r.stderr[-2000:])
    print(open(out, encoding="utf-8").read())
    return rc


if __name__ == "__main__":
    sys.exit(main())
