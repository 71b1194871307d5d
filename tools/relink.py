"""Recompile the named csrc/*.hip files only and relink lap_amd/liblap_hip.so (iteration helper: lap_amd.build rebuilds everything).
Writes the digest stamp, so that the next lap_amd.build.build() sees an up-to-date library."""
import subprocess, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
from lap_amd import build as B

objdir = B.ROOT / "build"
for name in sys.argv[1:]:
    subprocess.check_call(["/opt/rocm/bin/hipcc", *B.FLAGS, f"-I{objdir}", "-c", str(B.CSRC / name), "-o", str(objdir / (name + ".o"))])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *[str(objdir / (s + ".o")) for s in B.SOURCES], "-o", str(B.LIB)])
(B.ROOT / ".liblap_hip.digest").write_text(B._digest())
print("relinked", B.LIB)
