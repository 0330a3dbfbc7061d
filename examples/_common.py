"""Shared by the example scripts: make `import ins_amd` work from a checkout and parse `key=value` arguments."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cli(defaults):
    out = dict(defaults)
    for a in sys.argv[1:]:
        k, v = a.split("=", 1)
        out[k] = type(defaults[k])(v)
    return out
