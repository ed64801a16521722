"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/grut_amd.h declares."""
import importlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(grut_lib):
    abi = importlib.import_module("3dgrut_amd._abi")
    header = open(os.path.join(ROOT, "include", "grut_amd.h")).read()
    declared = set(re.findall(r"\b((?:gut|grt|grut)_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(abi.EXPORTED_SYMBOLS), declared ^ set(abi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(grut_lib, name), f"{name} missing from libgrut_amd.so"
    assert grut_lib.grut_abi_version() == abi.ABI_VERSION


def test_struct_sizes_match_header(tmp_path):
    """ctypes mirrors vs the C compiler's layout of include/grut_amd.h."""
    import ctypes
    import subprocess
    abi = importlib.import_module("3dgrut_amd._abi")
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "grut_amd.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(GrutCamera),sizeof(GutConfig),sizeof(GutFrame),sizeof(GutStats),sizeof(GrtConfig),sizeof(GrtFrame),sizeof(GrtStats));return 0;}')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mirrors = [abi.GrutCamera, abi.GutConfig, abi.GutFrame, abi.GutStats, abi.GrtConfig, abi.GrtFrame, abi.GrtStats]
    assert sizes == [ctypes.sizeof(m) for m in mirrors]


def test_missing_library_fails_loudly(tmp_path):
    abi = importlib.import_module("3dgrut_amd._abi")
    import pytest
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        abi.load_library(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "3dgrut_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle/" not in text, f


def test_primitive_codes_match_the_header_and_the_checker():
    """render.primitive_type names -> GRUT_PRIM_* codes: the plugin's table (3dgrut_amd/_abi.py), the C header's enum and the CPU checker's
    GrtConfig.primitive_type (oracle/grt_oracle.c: g_prim) must be one numbering - a shifted code would trace another proxy silently."""
    abi = importlib.import_module("3dgrut_amd._abi")
    header = open(os.path.join(ROOT, "include", "grut_amd.h")).read()
    enum = {name.lower(): int(val) for name, val in re.findall(r"GRUT_PRIM_([A-Z]+)\s*=\s*(\d+)", header)}
    assert enum == abi.GRT_PRIMITIVES, (enum, abi.GRT_PRIMITIVES)
    assert sorted(enum.values()) == list(range(len(enum)))
    checker = open(os.path.join(ROOT, "oracle", "grt_oracle.c")).read()
    for name in ("custom", "trisurfel", "trihexa", "sphere"):        # the codes the checker branches on by number
        assert re.search(rf"g_prim == {enum[name]}\b", checker), name
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_util
    assert parity_util.GRT_PRIMITIVE_CODES == enum
