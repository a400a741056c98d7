"""CPU: the C-ABI library loads and exports every symbol include/gpsat_hip.h declares; the ctypes
mirror of gpsat_batch has the C layout.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from gpsat_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "gpsat_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsat_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = _declared_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/gpsat_hip.h but not exported"
    assert sorted(L.EXPORTS) == names
    assert lib.gpsat_version() == L.ABI_VERSION


def test_header_constants_match_binding():
    src = open(HEADER).read()
    def const(name):
        return int(re.search(rf"#define\s+{name}\s+(-?\d+)", src).group(1))
    assert const("GPSAT_ABI_VERSION") == L.ABI_VERSION
    assert const("GPSAT_KERNEL_RBF") == L.KERNEL_IDS["RBF"]
    assert const("GPSAT_KERNEL_MATERN12") == L.KERNEL_IDS["Matern12"]
    assert const("GPSAT_KERNEL_MATERN32") == L.KERNEL_IDS["Matern32"]
    assert const("GPSAT_KERNEL_MATERN52") == L.KERNEL_IDS["Matern52"]
    assert (const("GPSAT_OPT_NONE"), const("GPSAT_OPT_LBFGS"), const("GPSAT_OPT_ADAM")) == (0, 1, 2)
    assert (const("GPSAT_MEM_HOST"), const("GPSAT_MEM_DEVICE")) == (L.MEM_HOST, L.MEM_DEVICE)


def test_struct_layout_matches_c(tmp_path):
    """Compile a tiny C program against the header and compare sizeof/offsetof with ctypes."""
    fields = [f[0] for f in L.GpsatBatch._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "gpsat_hip.h"', 'int main(){',
            'printf("%zu\\n", sizeof(gpsat_batch));']
    for f in fields:
        prog.append(f'printf("%zu\\n", offsetof(gpsat_batch, {f}));')
    prog.append('printf("%zu\\n", sizeof(gpsat_opts)); return 0;}')
    cfile = tmp_path / "layout.c"
    cfile.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)], check=True)
    vals = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert vals[0] == C.sizeof(L.GpsatBatch)
    for f, off in zip(fields, vals[1:-1]):
        assert getattr(L.GpsatBatch, f).offset == off, f
    assert vals[-1] == C.sizeof(L.GpsatOpts)


def test_no_device_is_reported_not_faked():
    """Without a GPU the library says so; the product path never falls back to the CPU."""
    lib = L.load()
    if lib.gpsat_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.gpsat_create(0, None, C.byref(h))
    assert rc != 0 and not h.value
    assert b"no HIP device" in lib.gpsat_last_error()
    from gpsat_amd.engine import Engine, GpsatError
    with pytest.raises(GpsatError):
        Engine(0)


def test_product_package_never_imports_oracle():
    """The oracle is test infrastructure: nothing under gpsat_amd/ may import it."""
    pkg = os.path.join(ROOT, "gpsat_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "gp_oracle" not in txt, f


def test_developer_knobs_are_gated():
    """No environment variable changes what the shipped library does unless GPSAT_DEVELOPER=1 is set: every GPSAT_DEBUG_*
    knob goes through dev_env(), and the only plain getenv in the native sources is the one that reads GPSAT_DEVELOPER."""
    src = os.path.join(ROOT, "gpsat_amd", "csrc")
    plain = []
    for f in os.listdir(src):
        if f.endswith((".cpp", ".hip", ".h")):
            for ln, line in enumerate(open(os.path.join(src, f)), 1):
                if "getenv(" in line and "dev_env(" not in line.split("getenv(")[0][-12:]:
                    plain.append((f, ln, line.strip()))
    assert all('"GPSAT_DEVELOPER"' in l or "return std::getenv(name)" in l for _, _, l in plain), plain
    assert len(plain) == 2, plain
