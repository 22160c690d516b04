"""Result egress parity: TableInfo::print / printall of include/aquery against the text the REFERENCE's own printer
produced for the same tables (tests/golden/print_shapes.txt, the stdout of oracle/_ref/print_shapes_ref = oracle/ref_harness.cpp
compiled with the reference headers over tests/emitted/print_shapes.inc).  Host-only: runs without a GPU."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
EM = os.path.join(HERE, "emitted")


def test_print_and_printall_text_is_the_reference_text_byte_for_byte():
    subprocess.check_call(["make", "-C", EM, "build/print_shapes"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(EM, "build", "print_shapes")], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr.decode()
    want = open(os.path.join(HERE, "golden", "print_shapes.txt"), "rb").read()
    assert out.stdout == want
    assert want.count(b"-- ") == 10 and b"1180591620717411303429" in want and b"133.333333" in want


def test_golden_text_is_what_the_reference_prints_where_it_is_built():
    ref_exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "print_shapes_ref")
    if not os.path.exists(ref_exe) or not os.path.isdir("/root/reference/server"):
        import pytest
        pytest.skip("reference not built here")
    got = subprocess.run([ref_exe], capture_output=True, timeout=120).stdout
    assert got == open(os.path.join(HERE, "golden", "print_shapes.txt"), "rb").read()
