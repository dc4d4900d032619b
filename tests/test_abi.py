"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/gdiet_hip.h declares.
No compute call is made here (there is no GPU in this container)."""
import ctypes
import os
import re

from conftest import ROOT, load_pkg


def _declared():
    text = open(os.path.join(ROOT, "include", "gdiet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdiet_hip_\w+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import sys
    sys.path.insert(0, os.path.join(ROOT, "genome-on-diet_amd"))
    import build as gbuild
    path = gbuild.build_hip()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "missing export " + n


def test_no_device_fails_loudly():
    """without a gfx950 device the context cannot be created -- there is no CPU fallback"""
    import torch
    if torch.cuda.is_available():
        return
    pkg = load_pkg()
    try:
        pkg.Context(0)
    except pkg.GdietError:
        return
    raise AssertionError("Context() must raise without a GPU")


def test_product_does_not_touch_oracle():
    """the product tree never imports / links / executes anything under oracle/"""
    bad = []
    for root, _, files in os.walk(os.path.join(ROOT, "genome-on-diet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".c")):
                s = open(os.path.join(root, f), errors="ignore").read()
                if re.search(r"\bgdo_|oracle/|import gdo|libgdo", s):
                    bad.append(f)
    assert not bad, bad
