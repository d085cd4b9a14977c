#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / LDS / scratch of every gfx950 kernel in a hipcc object file (from the code object's metadata).
   python tools/kernel_resources.py moonshine_amd/_build/k_attn.hip.o [name-filter]"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def main():
    obj = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    d = tempfile.mkdtemp()
    subprocess.run([LLVM + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, d + "/fb"], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + d + "/fb", "--output=" + d + "/co"], check=True)
    notes = subprocess.run([LLVM + "llvm-readelf", "--notes", d + "/co"], capture_output=True, text=True).stdout
    rec = {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "agpr_count" and rec.get("name"):
            pass
        if k in ("agpr_count", "group_segment_fixed_size", "private_segment_fixed_size", "sgpr_count", "vgpr_count",
                 "max_flat_workgroup_size", "name", "vgpr_spill_count"):
            if k == "agpr_count" and "agpr_count" in rec:   # a new kernel record starts (keys are sorted; agpr_count is first)
                emit(rec, flt)
                rec = {}
            rec[k] = v
    emit(rec, flt)


def emit(rec, flt):
    if not rec.get("name"):
        return
    name = subprocess.run(["c++filt", rec["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"msh::\(anonymous namespace\)::", "", name).split("(")[0]
    if flt and flt not in name:
        return
    print(f"{name[:84]:84s} vgpr {rec.get('vgpr_count', '?'):>3} agpr {rec.get('agpr_count', '?'):>3} sgpr {rec.get('sgpr_count', '?'):>3} "
          f"lds {rec.get('group_segment_fixed_size', '?'):>6} scratch {rec.get('private_segment_fixed_size', '?'):>4} "
          f"spill {rec.get('vgpr_spill_count', '0')}")


if __name__ == "__main__":
    main()
