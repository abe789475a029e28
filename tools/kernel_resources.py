#!/usr/bin/env python3
"""Per-kernel resource usage of the built gfx950 code objects, and the build's no-scratch gate.

    python tools/kernel_resources.py build/obj/*.o                  table: VGPRs, AGPRs, SGPRs, spills, scratch, LDS
    python tools/kernel_resources.py --check tools/hot_kernels.txt build/obj/*.o
        exit 1 when a kernel matching one of the patterns of the file uses scratch memory (private_segment_fixed_size
        > 0) or spills a register — `make engine` runs this after linking, so a hot kernel that starts to spill fails
        the build instead of waiting for somebody to read the ISA (round 3 shipped tp_expand_kernel<1024,3> with 24
        spilled VGPRs while the documentation said none).

Reads the AMDGPU metadata note (`llvm-readelf --notes`) of the device code object inside every host object
(section .hip_fatbin -> clang-offload-bundler).  No GPU needed.
"""
from __future__ import annotations

import argparse
import fnmatch
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = (
    ".vgpr_count",
    ".agpr_count",
    ".sgpr_count",
    ".vgpr_spill_count",
    ".sgpr_spill_count",
    ".private_segment_fixed_size",
    ".group_segment_fixed_size",
    ".max_flat_workgroup_size",
)


def demangle(names: list[str]) -> list[str]:
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return out.stdout.splitlines()


def kernels_of(obj: Path) -> list[dict]:
    with tempfile.TemporaryDirectory() as tmp:
        fat = Path(tmp) / "fat.bin"
        co = Path(tmp) / "dev.co"
        r = subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", str(obj), str(fat)], capture_output=True)
        if r.returncode != 0 or not fat.exists() or fat.stat().st_size == 0:
            return []
        r = subprocess.run(
            [str(LLVM / "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={fat}", f"--targets={TARGET}", f"--output={co}"],
            capture_output=True,
        )
        if r.returncode != 0:
            return []
        notes = subprocess.run([str(LLVM / "llvm-readelf"), "--notes", str(co)], capture_output=True, text=True, check=True).stdout
    out: list[dict] = []
    cur: dict | None = None
    in_kernels = False
    for line in notes.splitlines():
        if line.startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if in_kernels and re.match(r"^amdhsa\.\w+:", line):
            in_kernels = False
        if not in_kernels:
            continue
        m = re.match(r"^\s+(-\s+)?(\.[a-z_]+):\s*(.*)$", line)
        if not m:
            continue
        dash, key, value = m.groups()
        # a kernel's map starts at the "- " of indentation 2; argument maps are nested deeper
        indent = len(line) - len(line.lstrip())
        if dash and indent == 2:
            cur = {}
            out.append(cur)
        if cur is None or indent > 4:
            continue
        if key == ".name":
            cur["name"] = value.strip("'\"")
        elif key in FIELDS:
            cur[key] = int(value)
    out = [k for k in out if "name" in k]
    for k, d in zip(out, demangle([k["name"] for k in out])):
        k["demangled"] = re.sub(r"^void ", "", d)
        k["short"] = re.sub(r"\(.*$", "", re.sub(r"\(anonymous namespace\)::", "", k["demangled"])).replace("spmv::", "")
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("objects", nargs="+")
    ap.add_argument("--check", help="file of fnmatch patterns (one per line, # comments) over the short kernel names")
    ap.add_argument("--all", action="store_true", help="with --check: print every kernel, not only the matching ones")
    args = ap.parse_args()
    patterns: list[str] = []
    excluded: list[str] = []  # lines starting with "!": kernels no pattern shall claim (third-party code)
    if args.check:
        for line in Path(args.check).read_text().splitlines():
            line = line.split("#", 1)[0].strip()
            if line.startswith("!"):
                excluded.append(line[1:].strip())
            elif line:
                patterns.append(line)
    rows = []
    for obj in args.objects:
        for k in kernels_of(Path(obj)):
            k["object"] = Path(obj).name
            rows.append(k)
    bad = []
    matched = {p: 0 for p in patterns}
    print(f"{'kernel':<72} {'vgpr':>4} {'agpr':>4} {'sgpr':>4} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'lds':>6}")
    for k in sorted(rows, key=lambda k: (k["object"], k["short"])):
        hot = [] if any(fnmatch.fnmatch(k["short"], e) for e in excluded) else [p for p in patterns if fnmatch.fnmatch(k["short"], p)]
        for p in hot:
            matched[p] += 1
        if patterns and not hot and not args.all:
            continue
        scratch = k.get(".private_segment_fixed_size", 0)
        vs, ss = k.get(".vgpr_spill_count", 0), k.get(".sgpr_spill_count", 0)
        mark = ""
        if hot and (scratch > 0 or vs > 0):
            bad.append(k)
            mark = "  <-- HOT KERNEL USES SCRATCH"
        print(
            f"{k['short'][:72]:<72} {k.get('.vgpr_count', 0):>4} {k.get('.agpr_count', 0):>4} {k.get('.sgpr_count', 0):>4} "
            f"{vs:>6} {ss:>6} {scratch:>7} {k.get('.group_segment_fixed_size', 0):>6}{mark}"
        )
    if patterns:
        missing = [p for p, n in matched.items() if n == 0]
        if missing:
            print("patterns that matched no kernel (renamed? update the list):", ", ".join(missing), file=sys.stderr)
            return 1
        if bad:
            print(f"{len(bad)} hot kernel(s) use scratch memory or spill registers", file=sys.stderr)
            return 1
        print(f"no-scratch gate: {sum(matched.values())} hot kernel instances, none uses scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
