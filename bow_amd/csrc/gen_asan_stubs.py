#!/usr/bin/env python3
"""Stubs for the host-only sanitized build of libbowgpu (Makefile target `asan`): every bowgpu:: symbol the host translation units
reference but do not define (kernel launchers and size helpers living in the .hip files) becomes a function that aborts loudly."""
import subprocess
import sys

objs, out = sys.argv[1:-1], sys.argv[-1]
und = set(subprocess.check_output(["nm", "-u"] + objs, text=True).split())
dfn = {l.split()[-1] for l in subprocess.check_output(["nm", "--defined-only"] + objs, text=True).splitlines() if l.strip() and not l.endswith(":")}
syms = sorted(s for s in und if s.startswith("_ZN6bowgpu") and s not in dfn)
with open(out, "w") as f:
    f.write("#include <stdio.h>\n#include <stdlib.h>\n")
    for i, s in enumerate(syms):
        f.write('extern "C" void bowgpu_asan_stub_%d(void) __asm__("%s");\n' % (i, s))
        f.write('extern "C" void bowgpu_asan_stub_%d(void) { fprintf(stderr, "device-side symbol %s reached in the host-only sanitized build\\n"); abort(); }\n' % (i, s))
print("%d stubs" % len(syms))
