#!/usr/bin/env python3
"""check.py FILE.s : the EXEC-dependent instructions that stand in front of a join block's EXEC restore (csrc/hipcc_via_asm.py's lint)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "gaussian-ray-tracing_amd", "csrc"))
import hipcc_via_asm as V  # noqa: E402

bad = V.lint(open(sys.argv[1]).read())
for ln, t in bad:
    print(f"line {ln}: {t}")
sys.exit(1 if bad else 0)
