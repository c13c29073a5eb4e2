"""CPU: the companion filter mmannot-amd-addnh against the reference's addNH compiled from its source (oracle/_ref/addNH)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MINE = os.path.join(ROOT, "mmannot_amd", "mmannot-amd-addnh")
REF = os.path.join(ROOT, "oracle", "_ref", "addNH")

SAM = ("@HD\tVN:1.0\n@SQ\tSN:chr1\tLN:1000\n"
       "r1\t0\tchr1\t10\t255\t5M\t*\t0\t0\t*\t*\n"
       "r2\t0\tchr1\t20\t255\t5M\t*\t0\t0\t*\t*\n"
       "r2\t16\tchr1\t30\t255\t5M\t*\t0\t0\t*\t*\n"
       "\n"
       "r2\t0\tchr1\t40\t255\t5M\t*\t0\t0\t*\t*\n"
       "@CO\ta comment in the middle\n"
       "r3\t0\tchr1\t50\t255\t5M\t*\t0\t0\t*\t*\n"
       "r1\t0\tchr1\t60\t255\t5M\t*\t0\t0\t*\t*\n"
       "nofields\n")
EXPECTED = ("@HD\tVN:1.0\n@SQ\tSN:chr1\tLN:1000\n"
            "r1\t0\tchr1\t10\t255\t5M\t*\t0\t0\t*\t*\tNH:i:1\n"
            "\n"
            "@CO\ta comment in the middle\n"
            "r2\t0\tchr1\t20\t255\t5M\t*\t0\t0\t*\t*\tNH:i:3\n"
            "r2\t16\tchr1\t30\t255\t5M\t*\t0\t0\t*\t*\tNH:i:3\n"
            "r2\t0\tchr1\t40\t255\t5M\t*\t0\t0\t*\t*\tNH:i:3\n"
            "r3\t0\tchr1\t50\t255\t5M\t*\t0\t0\t*\t*\tNH:i:1\n"
            "r1\t0\tchr1\t60\t255\t5M\t*\t0\t0\t*\t*\tNH:i:1\n"
            "nofields\tNH:i:1\n")


def test_addnh_known_answer():
    out = subprocess.run([MINE], input=SAM.encode(), stdout=subprocess.PIPE, check=True).stdout.decode()
    assert out == EXPECTED
    assert subprocess.run([MINE, "-x"], stderr=subprocess.PIPE).returncode == 1


@pytest.mark.skipif(not os.path.exists(REF), reason="compiled reference only exists in the build container / travels with oracle/_ref")
def test_addnh_vs_reference(tmp_path):
    i = tmp_path / "in.sam"; i.write_text(SAM * 50)
    a = subprocess.run([MINE, "-i", str(i)], stdout=subprocess.PIPE, check=True).stdout
    b = subprocess.run([REF, "-i", str(i)], stdout=subprocess.PIPE, check=True).stdout
    assert a == b
