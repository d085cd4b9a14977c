"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` logs: one line per kernel."""
import re
import subprocess
import sys


def main(paths):
    for path in paths:
        txt = open(path).read()
        for b in re.split(r"(?=Function Name)", txt):
            m = re.search(r"Function Name: (\S+)", b)
            if not m:
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = name.replace("msh::(anonymous namespace)::", "")[:120]

            def g(k):
                mm = re.search(k + r": (\d+)", b)
                return mm.group(1) if mm else "?"

            print("%-120s V=%s A=%s spill=%s lds=%s occ=%s" % (name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))


if __name__ == "__main__":
    main(sys.argv[1:])
