#!/usr/bin/env python3
"""Undefined-global check for the GPU test modules (they cannot run in the CPU container, so a NameError would only show on the
GPU box): imports each module and resolves every global name its functions load against the module's globals and builtins."""
import builtins
import dis
import importlib
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def code_objects(co):
    yield co
    for c in co.co_consts:
        if isinstance(c, types.CodeType):
            yield from code_objects(c)


def main():
    bad = 0
    mods = sys.argv[1:] or [f[:-3] for f in sorted(os.listdir(os.path.join(ROOT, "tests"))) if f.startswith("test_") and f.endswith(".py")] + ["frame_check"]
    for name in mods:
        m = importlib.import_module(name)
        for fn in vars(m).values():
            if not isinstance(fn, types.FunctionType) or fn.__module__ != m.__name__:
                continue
            for co in code_objects(fn.__code__):
                for ins in dis.get_instructions(co):
                    if ins.opname in ("LOAD_GLOBAL", "LOAD_NAME") and ins.argval not in vars(m) and not hasattr(builtins, ins.argval):
                        print(f"{name}.{fn.__name__}: undefined global {ins.argval!r}")
                        bad += 1
    print("undefined globals:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
