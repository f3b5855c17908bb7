"""CPU: the C-ABI library loads and exports every symbol include/nsamd.h declares, and the ctypes binding lists
exactly those (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "nsamd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nsamd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert len(syms) >= 20 and "nsamd_hashgrid_encode_fwd" in syms and "nsamd_field_mlp_bwd" in syms


def test_library_exports_every_declared_symbol():
    from nerfstudio_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = ctypes.CDLL(_native.LIB_PATH)
    for sym in declared_symbols():
        assert hasattr(lib, sym), f"{sym} declared in include/nsamd.h but not exported by libnsamd.so"


def test_binding_matches_header():
    from nerfstudio_amd import _native

    assert sorted(_native._SIGNATURES) == declared_symbols()
    lib = _native.load()
    assert lib.nsamd_version().decode().startswith("nsamd")
    assert lib.nsamd_status_string(0).decode() == "ok"
    assert "not supported" in _native.status_string(-2)
    with pytest.raises(RuntimeError, match="status -1"):
        _native.check(-1, "unit test")


def test_argument_validation_without_gpu():
    """Entry points validate before launching: bad arguments return NSAMD_ERR_INVALID_ARG without touching HIP."""
    from nerfstudio_amd import _native as N

    lib = N.load()
    assert lib.nsamd_sh4_encode(None, 5, None, None) == -1
    assert lib.nsamd_sh4_encode(None, 0, None, None) == 0  # empty input is a no-op
    assert lib.nsamd_piecewise_bins(None, None, None, None, 0, 4, 0, 0, None, None, None) == -1
    g = N.make_grid(40 if False else 16, 19, [16.0] * 16)
    pts = N.make_points()
    assert lib.nsamd_hashgrid_encode_fwd(pts, 16, 0, N.Aabb(), None, g, None, 1, 16, None, None) == -1
    assert lib.nsamd_weights_fwd(None, None, 0, 8, None, None) == 0
    assert lib.nsamd_piecewise_bins(None, None, None, None, 0, 4, 8, 7, None, None, None) == -1  # unknown spacing mode
    # host-only workspace query: grows with M, the write-only variant adds the worst-case spill list, the zero-state
    # prefix (header + per-tile cursors) is a few KB
    g19 = N.make_grid(16, 19, [16.0 * 1.38**i for i in range(16)])
    assert lib.nsamd_hashgrid_encode_bwd_workspace(g19, 0, 0) == 0
    assert lib.nsamd_hashgrid_encode_bwd_workspace(g19, 100, 0) > 0
    w1, w2 = lib.nsamd_hashgrid_encode_bwd_workspace(g19, 196608, 0), lib.nsamd_hashgrid_encode_bwd_workspace(g19, 2 * 196608, 0)
    assert 0 < w1 < w2 and w1 * 4 < 2**31
    assert lib.nsamd_hashgrid_encode_bwd_workspace(g19, 196608, 1) > w1 + 2 * 196608 * 16 * 5  # + (4 - 1) M L spill records of 20 B
    state = lib.nsamd_hashgrid_encode_bwd_workspace_state(g19, 196608)
    assert 64 < state <= 64 + 16 * 64 + 4 and state % 4 == 0
    assert lib.nsamd_linear_fwd(None, None, None, 4, 0, 3, 0, None, None) == -1


def test_header_is_plain_c_and_a_c_program_links_against_the_library(tmp_path):
    """The boundary is a C ABI: include/nsamd.h must compile as C99 (no C++ / torch types), and a C translation unit that
    references every declared entry point must link against libnsamd.so (host-only calls: version, status strings,
    argument validation — no GPU needed)."""
    import subprocess

    from nerfstudio_amd import _native

    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", HEADER], check=True)
    src = tmp_path / "use.c"
    refs = "\n".join(f"    (void*)&{s}," for s in declared_symbols())
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "nsamd.h"\n'
        "static void* const table[] = {\n" + refs + "\n};\n"
        "int main(void) {\n"
        '  if (strncmp(nsamd_version(), "nsamd", 5) != 0) return 1;\n'
        '  if (strcmp(nsamd_status_string(0), "ok") != 0) return 2;\n'
        "  if (nsamd_sh4_encode(NULL, 5, NULL, NULL) != -1) return 3;   /* invalid argument, nothing launched */\n"
        "  if (nsamd_sh4_encode(NULL, 0, NULL, NULL) != 0) return 4;    /* empty input is a no-op */\n"
        '  printf("%d symbols\\n", (int)(sizeof(table) / sizeof(table[0])));\n'
        "  return 0;\n}\n")
    exe = tmp_path / "use"
    libdir = os.path.dirname(_native.LIB_PATH)
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe), "-L", libdir, "-lnsamd",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.strip() == f"{len(declared_symbols())} symbols"
