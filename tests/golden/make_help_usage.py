#!/usr/bin/env python3
"""Extract the one golden text the reference holds: the output of `pansim --help` of a real Pansim binary (clap 3), pasted
into its README (/root/reference/README.md:40-138, the fenced block behind "All command line arguments can be found below").
The lines from `USAGE:` on are written to tests/golden/help_usage.txt -- data (an expected output), cited here with its
source; tests/test_cli.py compares `pansim --help` with it byte for byte.  Run where /root/reference exists:
    python tests/golden/make_help_usage.py [README.md]"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
readme = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/README.md"
lines = open(readme).read().split("\n")
start = next(i for i, line in enumerate(lines) if line.startswith("All command line arguments can be found below"))
first = next(i for i in range(start, len(lines)) if lines[i].strip() == "```") + 1
last = next(i for i in range(first, len(lines)) if lines[i].strip() == "```")
block = lines[first:last]
assert block[0] == "USAGE:", block[0]
open(os.path.join(HERE, "help_usage.txt"), "w").write("\n".join(block) + "\n")
print("wrote %d lines from %s:%d-%d" % (len(block), readme, first + 1, last))
