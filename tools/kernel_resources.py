#!/usr/bin/env python3
"""Static resources of every gfx950 kernel in the built library, read from the code objects (no GPU needed).

For each object file of smoqyelphqmc.jl_amd/csrc the gfx950 code object is taken out of its .hip_fatbin section
(llvm-objcopy + clang-offload-bundler) and the AMDGPU metadata note is read (llvm-readelf --notes): VGPRs, AGPRs, SGPRs,
static LDS, scratch ("private segment") and spill counts per kernel.  Wavefronts per SIMD follow from the unified
512-register file of a CDNA4 SIMD lane (allocation granule 8, at most 8 wavefronts), which is what the occupancy
statements of DESIGN.md §8 and docs/DESIGN_LOG.md rest on.

    python tools/kernel_resources.py            # table on stdout (profiles/r04_kernel_resources.txt is this output)
    python tools/kernel_resources.py --json     # the same as JSON

tests/test_kernel_resources.py asserts on it: no scratch in the kernels of a CG iteration, and the register budgets the
design quotes.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "smoqyelphqmc.jl_amd", "csrc")
LLVM = os.environ.get("SMOQY_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

_FIELDS = {
    ".name": "symbol",
    ".vgpr_count": "vgpr",
    ".agpr_count": "agpr",
    ".sgpr_count": "sgpr",
    ".group_segment_fixed_size": "lds",
    ".private_segment_fixed_size": "scratch",
    ".vgpr_spill_count": "vgpr_spill",
    ".sgpr_spill_count": "sgpr_spill",
    ".max_flat_workgroup_size": "max_wg",
    ".uses_dynamic_stack": "dyn_stack",
}


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def waves_per_simd(vgpr, agpr=0):
    """Wavefronts one SIMD holds at this register count: 512 unified registers per lane, granule 8, at most 8."""
    regs = max(8, (vgpr + agpr + 7) // 8 * 8)
    return min(8, 512 // regs)


def _kernels_of_note(text):
    """Kernel records of one `llvm-readelf --notes` dump (the metadata is YAML; a line scan is enough for the scalars)."""
    out, cur, in_kernels = [], None, False
    for line in text.splitlines():
        s = line.strip()
        if s.startswith("amdhsa.kernels:"):
            in_kernels = True
            continue
        if s.startswith("amdhsa.target:") or s.startswith("amdhsa.version:"):
            in_kernels = False
        if not in_kernels:
            continue
        if line.startswith("  - ."):                        # a kernel record opens at indent 2; its scalars sit at indent 4
            cur = {}
            out.append(cur)
            s = s[2:]
        elif not (line.startswith("    .") and cur is not None):
            continue                                       # argument records and other nested maps
        m = re.match(r"(\.[a-z_]+):\s+(.*)$", s)
        if m and m.group(1) in _FIELDS and _FIELDS[m.group(1)] not in cur:
            v = m.group(2).strip().strip("'\"")
            k = _FIELDS[m.group(1)]
            cur[k] = v if k in ("symbol", "dyn_stack") else int(v)
    return [k for k in out if "symbol" in k and "vgpr" in k]


def collect(csrc=CSRC):
    objs = sorted(f for f in os.listdir(csrc) if f.endswith(".o"))
    if not objs:
        raise SystemExit("no object files under %s: run `make -C smoqyelphqmc.jl_amd/csrc` first" % csrc)
    kernels = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            fat, co = os.path.join(tmp, o + ".fat"), os.path.join(tmp, o + ".co")
            try:
                _run(os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, os.path.join(csrc, o))
            except subprocess.CalledProcessError:
                continue                                   # a host-only object (member.o)
            if not os.path.exists(fat) or os.path.getsize(fat) == 0:
                continue
            _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                 "--targets=" + TARGET, "--output=" + co)
            recs = _kernels_of_note(_run(os.path.join(LLVM, "llvm-readelf"), "--notes", co))
            if recs:
                names = _run("c++filt", *[r["symbol"] for r in recs]).splitlines()
                for r, n in zip(recs, names):
                    r["unit"] = o[:-2]
                    r["kernel"] = _short(n)
                    r.setdefault("agpr", 0)
                    r["waves_per_simd"] = waves_per_simd(r["vgpr"], r["agpr"])
                kernels += recs
    return kernels


def _short(demangled):
    """`void smoqy::(anonymous namespace)::cheb_wave_kernel<3>(smoqy::KpmArgs, …)` -> `cheb_wave_kernel<3>`."""
    s = demangled
    if s.startswith("void "):
        s = s[5:]
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):                              # the argument list opens at the first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not s.startswith("(anonymous namespace)", i):
            cut = i
            break
    s = s[:cut]
    return s.replace("smoqy::", "").replace("(anonymous namespace)::", "")


def main(argv):
    ks = collect()
    if "--json" in argv:
        print(json.dumps(ks, indent=1))
        return 0
    print("# gfx950 kernel resources of libsmoqy_hip.so, from the code objects (tools/kernel_resources.py; no GPU involved)")
    print("# waves/SIMD = min(8, 512 // roundup8(VGPR + AGPR)); scratch = private segment bytes per lane")
    print("%-20s %-72s %5s %5s %5s %7s %7s %6s %6s %5s" % ("unit", "kernel", "VGPR", "AGPR", "SGPR", "LDS", "scratch", "vspill", "sspill",
                                                          "w/SIMD"))
    for k in sorted(ks, key=lambda k: (k["unit"], k["kernel"])):
        print("%-20s %-72s %5d %5d %5d %7d %7d %6d %6d %5d" % (k["unit"], k["kernel"][:72], k["vgpr"], k["agpr"], k["sgpr"],
                                                              k.get("lds", 0), k.get("scratch", 0), k.get("vgpr_spill", 0),
                                                              k.get("sgpr_spill", 0), k["waves_per_simd"]))
    n_scr = sum(1 for k in ks if k.get("scratch", 0) or k.get("vgpr_spill", 0))
    print("# %d kernels, %d with scratch or VGPR spills" % (len(ks), n_scr))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
