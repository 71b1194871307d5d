"""CPU-side checks of the C-ABI boundary: the library builds, loads, and exports exactly the
entry points include/lap_hip.h declares (no compute calls: there is no GPU here)."""
import ctypes
import pathlib
import re

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _header_names():
    h = (ROOT / "include" / "lap_hip.h").read_text()
    return set(re.findall(r"\bint (lap_[a-z0-9_]+)\(", h))


def test_library_exports_every_declared_symbol():
    from lap_amd.build import build

    lib = ctypes.CDLL(str(build(verbose=False)))
    names = _header_names()
    assert len(names) >= 37
    for n in names:
        assert hasattr(lib, n), f"{n} declared in lap_hip.h but not exported"
    assert lib.lap_abi_version() == 1


def test_binding_covers_header():
    src = (ROOT / "lap_amd" / "hip.py").read_text()
    bound = set(re.findall(r'"(lap_[a-z0-9_]+)":', src))
    assert bound == _header_names()


def test_assembly_kernels_file_is_the_generators_output():
    """lap_amd/csrc/gemm_asm_kernels.s is generated (tools/gen_gemm_asm.py): the committed file must be what the generator
    writes with default schedule parameters, so that nobody edits the 16k lines by hand."""
    import os
    import pathlib
    import subprocess
    import sys

    root = pathlib.Path(__file__).resolve().parents[1]
    env = {k: v for k, v in os.environ.items() if not k.startswith("ASM_")}
    out = subprocess.run([sys.executable, str(root / "tools" / "gen_gemm_asm.py")], capture_output=True, text=True, check=True, env=env).stdout
    assert out == (root / "lap_amd" / "csrc" / "gemm_asm_kernels.s").read_text()
