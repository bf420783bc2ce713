#!/usr/bin/env python
"""Static resources of every kernel in libgpd_hip.so, read from the gfx950 code objects the Makefile's .o files carry
(no GPU needed): VGPR / AGPR / SGPR counts, LDS bytes, scratch bytes, spills, the waves per SIMD those allow at the
kernel's launch size, and how many matrix instructions of which kind the body holds.

    python profiles/isa_stats.py > profiles/r06_isa_stats.txt        (after `make` in gpd_amd/csrc)

tests/test_isa_resources.py reads the same numbers: no kernel of the scoring path spills or touches scratch, the
LDS budgets are the ones DESIGN.md states.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpd_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
UNITS = ("context", "lenet", "lenet_fast", "search", "images", "plan", "preprocess", "cluster")


def _run(*a):
    return subprocess.run(a, check=True, capture_output=True, text=True).stdout


def code_object(unit, tmp):
    """the gfx950 ELF inside <unit>.o (None when the unit has no device code)"""
    obj = os.path.join(CSRC, unit + ".o")
    if not os.path.exists(obj):
        raise FileNotFoundError(obj + " (run make in gpd_amd/csrc)")
    fat = os.path.join(tmp, unit + ".fatbin")
    subprocess.run([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    co = os.path.join(tmp, unit + ".co")
    subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + fat, "--output=" + co], check=True)
    return co


def metadata(co):
    notes = _run(LLVM + "/llvm-readelf", "--notes", co)
    y = notes[notes.index("---"):]
    y = y[:y.index("\n...")] if "\n..." in y else y
    return yaml.safe_load(y)


MATRIX = re.compile(r"\b(v_mfma_\w+|v_smfmac_\w+)")


def matrix_instructions(co):
    """{kernel symbol: Counter of MFMA mnemonics in its body}"""
    dis = _run(LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co)
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), collections.Counter())
            continue
        if cur is not None:
            m = MATRIX.search(line)
            if m:
                cur[m.group(1)] += 1
    return out


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return p.stdout.splitlines()


def short(name):
    """void ns::kernel<A, B>(args...) -> kernel<A, B>"""
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return name[:cut]


def waves_per_simd(vgpr, agpr, lds, wg_threads):
    """gfx950: 512 unified registers per lane and SIMD in blocks of 8, at most 8 waves per SIMD; 160 KB of LDS per CU over
    whole workgroups (4 SIMDs share them)."""
    regs = -(-(vgpr + agpr) // 8) * 8 if agpr else -(-vgpr // 8) * 8
    by_regs = min(8, 512 // max(regs, 8))
    waves_wg = max(1, -(-wg_threads // 64))
    wgs_by_regs = by_regs * 4 // waves_wg
    wgs_by_lds = (160 * 1024) // lds if lds else 1 << 30
    wgs = max(0, min(wgs_by_regs, wgs_by_lds))
    return by_regs, wgs, wgs * waves_wg / 4.0


def collect():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for unit in UNITS:
            co = code_object(unit, tmp)
            if co is None:
                continue
            md = metadata(co)
            mi = matrix_instructions(co)
            ks = md.get("amdhsa.kernels", [])
            names = demangle([k[".name"] for k in ks])
            for k, nm in zip(ks, names):
                vg, ag = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
                lds = int(k.get(".group_segment_fixed_size", 0))
                wg = int(k.get(".max_flat_workgroup_size", 1024))
                rows.append(dict(unit=unit, kernel=short(nm), symbol=k[".name"], vgpr=vg, agpr=ag, sgpr=int(k.get(".sgpr_count", 0)), lds=lds,
                                 scratch=int(k.get(".private_segment_fixed_size", 0)), vgpr_spills=int(k.get(".vgpr_spill_count", 0)),
                                 sgpr_spills=int(k.get(".sgpr_spill_count", 0)), wg=wg, dynamic_stack=bool(k.get(".uses_dynamic_stack", False)),
                                 matrix=dict(mi.get(k[".name"], {}))))
    return rows


def main():
    rows = collect()
    print("# kernels of libgpd_hip.so (gfx950 code objects of gpd_amd/csrc/*.o; profiles/isa_stats.py).  wg = the launch bound the kernel was")
    print("# compiled for (__launch_bounds__, 1024 when none); waves/SIMD = what registers and STATIC LDS allow at that size, 8 at most (kernels")
    print("# launched with dynamic LDS — neighbourhood_kernel: 72 KB, hand_eval_kernel — hold fewer).  spills v / s = VGPR spills (to scratch")
    print("# memory) / SGPR spills (to lanes of a VGPR: no memory traffic).\n")
    print("%-12s %-58s %5s %5s %5s %8s %8s %9s %5s %10s  %s" % ("unit", "kernel", "vgpr", "agpr", "sgpr", "LDS B", "scratch", "spill v/s", "wg", "waves/SIMD", "matrix instructions"))
    for r in rows:
        _, wgs, wps = waves_per_simd(r["vgpr"], r["agpr"], r["lds"], r["wg"])
        mi = ", ".join("%d x %s" % (n, m) for m, n in sorted(r["matrix"].items()))
        print("%-12s %-58s %5d %5d %5d %8d %8d %9s %5d %10.1f  %s"
              % (r["unit"], r["kernel"][:58], r["vgpr"], r["agpr"], r["sgpr"], r["lds"], r["scratch"], "%d/%d" % (r["vgpr_spills"], r["sgpr_spills"]), r["wg"], wps, mi))
    print("\n%d kernels; %d with scratch memory, %d with VGPR spills, %d with SGPR spills"
          % (len(rows), sum(1 for r in rows if r["scratch"]), sum(1 for r in rows if r["vgpr_spills"]), sum(1 for r in rows if r["sgpr_spills"])))


if __name__ == "__main__":
    sys.exit(main())
